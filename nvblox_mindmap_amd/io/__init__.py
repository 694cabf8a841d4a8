"""On-disk / wire formats of the path: map checkpoints, the dataset's vertex-feature, depth,
pose and intrinsics files."""
