"""bench.py's legs (bench.py keeps the contract: ranks, the headline region, the record and its compact last line)."""
