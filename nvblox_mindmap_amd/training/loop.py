"""The iteration-based training loop and its evaluation pass (the flow of mindmap/run_training.py:262-770 without its
control plane: no Tap arguments, wandb, embodiments or visualiser -- the caller hands over loaders, model and optimizer).

What is mirrored, because it decides what the GPUs and RCCL see:
  * iterations, not epochs: a fresh loader iterator at every epoch boundary, ``sampler.set_epoch`` only every 5th epoch and
    only when world_size > 1 (:653-661); the validation sampler is shuffled once (``set_epoch(0)``, :631-632);
  * one optimizer step and one LinearLR step per iteration (:667-683);
  * every ``val_freq`` iterations: evaluation on ``num_batches`` batches in inference mode (full reverse diffusion), the per-rank
    means ALL-GATHERED and averaged over ranks (:371-375, synchronize_between_processes :443-452) -- the eval-time collective
    of SURVEY 8(e) --, then rank 0 writes last.pth / best.pth (:747-752) and every rank synchronises (:753)."""
from typing import Callable, Dict, Optional

import torch

from ..diffuser_actor.loss import compute_metrics
from ..nvblox_torch.timer import Timer
from .checkpoint import save_checkpoint
from .distributed import all_gather_objects, barrier, get_rank, get_world_size
from .trainer import build_lr_scheduler, train_one_step, unpack_batch

_MEANS = {"distance_m": "mean_distance_m", "distance_m_std": "distance_m_std", "rot_l1": "mean_rot_l1", "rot_error_deg": "mean_rot_error_deg",
          "openness_l1": "mean_openness_l1", "distance_m_x": "mean_distance_m_x", "distance_m_std_x": "distance_m_std_x",
          "distance_m_y": "mean_distance_m_y", "distance_m_std_y": "distance_m_std_y", "distance_m_z": "mean_distance_m_z",
          "distance_m_std_z": "distance_m_std_z", "head_yaw_error_deg": "mean_head_yaw_error_deg"}


@torch.no_grad()
def evaluate_nsteps(cfg, model, loader, num_batches: int = -1, unpack: Callable = unpack_batch) -> Dict[str, float]:
    """Mean losses and proxy metrics over ``num_batches`` batches of ``loader`` (-1: all), each batch run in INFERENCE mode (the
    full reverse-diffusion loop, as the reference evaluates), averaged over batches on every rank and then over ranks
    (run_training.py:262-375).  Names as the reference logs them (mean_total_loss, mean_distance_m, mean_bias_x, ...).
    Leaves the model in eval mode, like the reference; ``run_training`` switches back."""
    if num_batches == -1:
        n = len(loader)
    elif num_batches > 0:
        n = min(num_batches, len(loader))
    else:
        raise ValueError("Number of batches for evaluation shall be -1 or greater than 0.")
    model.eval()
    acc: Dict[str, torch.Tensor] = {}

    def add(name, value):
        acc[name] = acc.get(name, 0.0) + value.detach().to(torch.float64) / n

    for i, batch in enumerate(loader):
        if i == n:
            break
        s = unpack(cfg, batch)
        with Timer("step/eval/inference"):
            pred, head_yaw_pred, losses, _, _ = model(s["gt_gripper_pred"], s["gt_head_yaw"], s["rgbs"], s["pcds"], s["pcd_valid_mask"],
                                                      s["vertex_features"], s["vertices"], s["vertices_valid_mask"], None,
                                                      s["gripper_history"], run_inference=True)
        assert pred.shape == s["gt_gripper_pred"].shape
        for name, value in zip(("mean_total_loss", "mean_pos_loss", "mean_rot_loss", "mean_gripper_loss", "mean_head_yaw_loss"), losses):
            if value is not None and (name != "mean_head_yaw_loss" or cfg.predict_head_yaw):
                add(name, value)
        m = compute_metrics(pred, head_yaw_pred, s["gt_gripper_pred"], s["gt_head_yaw"], cfg.predict_head_yaw)
        for key, name in _MEANS.items():
            if key in m:
                add(name, m[key])
        for k, axis in enumerate("xyz"):
            add(f"mean_bias_{axis}", m["bias"][k])
    mine = {k: float(v) for k, v in acc.items()}  # (one host read per value: evaluation is not the hot loop)
    everyone = all_gather_objects(mine)
    return {k: sum(d[k] for d in everyone) / len(everyone) for k in mine}


def run_training(cfg, model, optimizer, train_loader, validation_loader, train_iters: int, val_freq: int, train_sampler=None,
                 validation_sampler=None, start_iter: int = 0, best_loss: Optional[float] = None, checkpoint_dir: Optional[str] = None,
                 num_batches_per_test_eval: int = -1, num_batches_per_train_eval: int = 0, lr_scheduler=None,
                 learning_rate_convergence_percentage: float = 0.75, learning_rate_end_factor: float = 0.5,
                 on_eval: Optional[Callable[[int, str, Dict[str, float]], None]] = None, unpack: Callable = unpack_batch,
                 graphed=None):
    """Iterations ``start_iter .. train_iters - 1`` (resume: pass what ``load_train_checkpoint(..., initial_learning_rate=)``
    returned).  ``model`` is the
    (DDP-wrapped) policy.  Returns (iterations done, best validation loss).  ``on_eval(step_id, split, values)`` receives every
    evaluation result (the reference logs them to wandb); ``num_batches_per_train_eval`` > 0 or -1 also evaluates on the
    training loader first, like ``skip_train_val=False``.

    ``graphed``: a ``training.GraphedTrainStep`` built on ``model`` (the BARE policy, not DDP-wrapped): the iteration's step is
    then its captured form -- flat gradient buffer, one explicit all-reduce, HIP graphs on a GPU -- instead of ``train_one_step``
    on a DDP wrapper; ``optimizer`` is ignored for stepping (pass ``graphed`` itself: it has the ``state_dict`` /
    ``load_state_dict`` the checkpoints use) and the LinearLR ramp is the step's own (``graphed.linear_lr``).  The loop reads one
    batch ahead so that the step can run the next batch's frozen backbone beside the current batch's trainable pass."""
    epoch_len = len(train_loader)
    assert epoch_len != 0, "Train loader contains less than one batch."
    assert len(validation_loader) != 0, "Validation loader contains less than one batch."
    if graphed is not None:
        graphed.linear_lr(train_iters, learning_rate_convergence_percentage, learning_rate_end_factor)
    elif lr_scheduler is None:
        # (a resumed run restarts the ramp from the initial learning rate, like the reference: load_train_checkpoint resets the
        # rate, checkpoint.py:124-127, and :603-611 builds a fresh LinearLR)
        lr_scheduler = build_lr_scheduler(optimizer, train_iters, learning_rate_convergence_percentage, learning_rate_end_factor)
    if validation_sampler is not None and get_world_size() > 1:
        validation_sampler.set_epoch(0)  # evaluation batches keep one order for the whole run
    model.train()
    it = None
    step_id = start_iter - 1
    ahead = None  # (graphed) the batch read one iteration early

    def next_batch(at_step):
        """The batch of iteration `at_step` (None past the end): a fresh iterator at every epoch boundary, as the reference does."""
        nonlocal it
        if at_step >= train_iters:
            return None
        if at_step % epoch_len == 0 or at_step == start_iter:
            if train_sampler is not None and get_world_size() > 1 and (at_step // epoch_len) % 5 == 0:
                train_sampler.set_epoch(at_step // epoch_len)
            it = iter(train_loader)
        return next(it)

    for step_id in range(start_iter, train_iters):
        with Timer("step"):
            with Timer("step/load_batch"):
                batch = ahead if ahead is not None else next_batch(step_id)
                ahead = None
                # One batch ahead for the captured step -- but not across an evaluation: it re-reads the training loader (a second
                # live iterator of the same workers), and the reference's loop starts its next iterator only afterwards.
                if graphed is not None and (step_id + 1) % val_freq != 0 and (step_id + 1) % epoch_len != 0:
                    ahead = next_batch(step_id + 1)
            with Timer("step/train"):
                if graphed is not None:
                    graphed.step(batch, ahead)
                else:
                    train_one_step(cfg, model, optimizer, batch, unpack=unpack)
            if graphed is not None:
                graphed.scheduler_step()
            else:
                lr_scheduler.step()
            if (step_id + 1) % val_freq == 0:
                if num_batches_per_train_eval:
                    with Timer("step/eval/train-val"):
                        values = evaluate_nsteps(cfg, model, train_loader, num_batches_per_train_eval, unpack)
                    if on_eval is not None:
                        on_eval(step_id, "train-val", values)
                with Timer("step/eval/val"):
                    values = evaluate_nsteps(cfg, model, validation_loader, num_batches_per_test_eval, unpack)
                if on_eval is not None:
                    on_eval(step_id, "val", values)
                if get_rank() == 0 and checkpoint_dir is not None:
                    best_loss = save_checkpoint(checkpoint_dir, model, graphed if graphed is not None else optimizer, step_id,
                                                values["mean_total_loss"], best_loss)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                barrier()  # nobody trains on while rank 0 is still writing the checkpoint another rank may resume from
                model.train()
    return step_id + 1, best_loss
