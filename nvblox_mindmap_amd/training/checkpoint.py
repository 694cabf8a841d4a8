"""Training checkpoints (mirror of mindmap/model_utils/checkpoint.py:30-52,103-136): ``last.pth`` after every validation
round, ``best.pth`` when the validation loss improved; each holds {weight, optimizer, iter, best_loss} -- the reference's
container layout.  "weight" is the state_dict of whatever wraps the model (a DDP wrapper saves "module."-prefixed names,
which the loaders accept by stripping / adding the prefix).

The parameter NAMES inside "weight" differ between the two code bases (this package's modules are organised differently);
a checkpoint written by the reference is recognised by its key names and converted on load
(diffuser_actor/reference_weights.py, pinned by tests/test_cpu_policy_golden.py); ``export_reference_checkpoint`` writes the
reference's names.  Optimizer state is NOT interchangeable (parameter order and grouping differ): loading a reference
checkpoint for training restores the weights and restarts the optimizer."""
import os
import warnings
import pathlib
from typing import Optional, Tuple

import torch


def save_checkpoint(checkpoint_log_dir, model, optimizer, step_id: int, new_loss: Optional[float], best_loss: Optional[float]):
    """Write last.pth (always) and best.pth (if new_loss <= best_loss or either is None); returns the updated best loss.
    Call on rank 0 only (run_training.py:747-752).  Files appear atomically (written to .tmp, then renamed)."""
    os.makedirs(checkpoint_log_dir, exist_ok=True)

    def dump(name, best):
        path = pathlib.Path(checkpoint_log_dir) / name
        tmp = str(path) + ".tmp"
        torch.save({"weight": model.state_dict(), "optimizer": optimizer.state_dict(), "iter": step_id + 1, "best_loss": best}, tmp)
        os.replace(tmp, path)

    if new_loss is None or best_loss is None or new_loss <= best_loss:
        best_loss = new_loss
        dump("best.pth", best_loss)
    dump("last.pth", best_loss)
    return best_loss


def _match_prefix(state: dict, model) -> dict:
    want = any(k.startswith("module.") for k in model.state_dict())
    have = any(k.startswith("module.") for k in state)
    if want == have:
        return state
    return {("module." + k) if want else k[len("module."):]: v for k, v in state.items()}


def _load_weights(model, state: dict) -> bool:
    """Returns True if `state` was a reference-named state dict (converted on the way in)."""
    from ..diffuser_actor import reference_weights as RW

    if RW.is_reference_state_dict(state):
        RW.load_reference_state_dict(model, state)
        return True
    state = _match_prefix(state, model)
    # checkpoints written before the trajectory-language attention existed (use_instruction models only use it) lack its
    # parameters: they keep their initial values; any OTHER missing or unexpected key is an error
    result = model.load_state_dict(state, strict=False)
    optional = ("traj_lang_attention", "traj_lang_ffn")
    missing = [k for k in result.missing_keys if not any(o in k for o in optional)]
    if missing or result.unexpected_keys:
        raise RuntimeError(f"checkpoint does not fit the model: missing {missing}, unexpected {list(result.unexpected_keys)}")
    if result.missing_keys:
        warnings.warn(f"checkpoint has no trajectory-language attention weights ({len(result.missing_keys)} tensors keep their "
                      "initial values); irrelevant unless use_instruction is set")
    return False


def load_inference_checkpoint(checkpoint_path: str, model, device):
    """mindmap/model_utils/checkpoint.py:103-114; accepts checkpoints of either code base."""
    assert checkpoint_path is not None and os.path.exists(checkpoint_path), checkpoint_path
    model_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    _load_weights(model, model_dict["weight"])
    model.eval()
    return model.to(device=device)


def export_reference_checkpoint(path: str, model, step_id: int = 0, best_loss: Optional[float] = None) -> None:
    """Write {weight, iter, best_loss} with the REFERENCE's parameter names and layouts, loadable by the reference's
    load_inference_checkpoint (no optimizer entry: its load_train_checkpoint treats it as optional, :124)."""
    from ..diffuser_actor.reference_weights import to_reference_state_dict

    torch.save({"weight": to_reference_state_dict(model.state_dict()), "iter": step_id + 1, "best_loss": best_loss}, path)


def load_train_checkpoint(checkpoint_path: str, model, optimizer, initial_learning_rate: Optional[float] = None) -> Tuple[int, Optional[float]]:
    """Restore weights + optimizer state, reset the learning rate (the scheduler restarts, checkpoint.py:124-127); returns
    (start_iter, best_loss)."""
    assert checkpoint_path is not None and os.path.exists(checkpoint_path), checkpoint_path
    model_dict = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    from_reference = _load_weights(model, model_dict["weight"])
    if "optimizer" in model_dict and not from_reference:
        saved = model_dict["optimizer"]
        flat_saved, flat_step = isinstance(saved, dict) and saved.get("format") == "flat-v1" or (isinstance(saved, dict) and "layout" in saved), hasattr(optimizer, "refresh_frozen_weights")
        if flat_saved != flat_step:
            # both loops write last.pth / best.pth: a checkpoint of one kind of step cannot restore the optimizer state of the other
            raise ValueError(f"{checkpoint_path}: optimizer state written by {'training.GraphedTrainStep' if flat_saved else 'a per-parameter AdamW'}, "
                             f"this run steps with {'training.GraphedTrainStep' if flat_step else 'a per-parameter AdamW'}: resume with the same kind of step "
                             "(run_training(graphed=...)), or load the weights only")
        optimizer.load_state_dict(saved)
        if flat_step:
            optimizer.refresh_frozen_weights()  # a captured backbone graph multiplies by split copies of the weights just loaded
        if initial_learning_rate is not None:
            if hasattr(optimizer, "set_lr"):  # a training.GraphedTrainStep (its rate lives in one device scalar)
                optimizer.set_lr(initial_learning_rate)
            else:
                for g in optimizer.param_groups:
                    g["lr"] = initial_learning_rate
    return model_dict.get("iter", 0), model_dict.get("best_loss", None)
