"""Restatement of the loop body of the reference's `Trainer.train_epoch` (engine/training_engine.py:221-300) — TEST INFRASTRUCTURE.

The GPU box has no reference tree, so the GPU test of the drop-in boundary cannot instantiate the reference `Trainer`; it drives the
class-swapped model through THIS loop instead.  The loop is pinned on the CPU box: tests/test_launch_cpu.py trains the same reference-built
model once with the reference's own `Trainer` (through cvnets_amd.launch) and once with `train_iterations`, and requires bit-identical
parameters.  Duck-typed like the engine: `criterion(input_sample=, prediction=, target=, epoch=, iterations=)`, `scheduler.update_lr(optimizer=,
epoch=, curr_iter=)`, `model_ema.update_parameters(model)`.
"""
import contextlib

import torch


def train_iterations(model, criterion, optimizer, scheduler, gradient_scaler, batches, *, device, epoch=0, start_iteration=0, accum_freq=1,
                     max_norm=None, amp_dtype=None, model_ema=None, set_to_none=True):
    """returns (number of optimizer updates, list of per-batch loss values)"""
    model.train()                                                           # :215
    criterion.train()                                                       # :218
    optimizer.zero_grad(set_to_none=set_to_none)                            # :224 _zero_grad
    iters, losses = start_iteration, []
    for batch_id, batch in enumerate(batches):                              # :229
        samples = batch["samples"].to(device, non_blocking=True)           # :235 move_to_device
        targets = batch["targets"].to(device, non_blocking=True)
        optimizer = scheduler.update_lr(optimizer=optimizer, epoch=epoch, curr_iter=iters)  # :246-248
        ctx = torch.autocast(device_type=torch.device(device).type, dtype=amp_dtype) if amp_dtype is not None else contextlib.nullcontext()
        with ctx:                                                           # :257-260 autocast_fn
            pred = model(samples)                                           # :262
            loss = criterion(input_sample=samples, prediction=pred, target=targets, epoch=epoch, iterations=iters)  # :264-270
            if isinstance(loss, dict):
                loss = loss["total_loss"]                                   # :272-277
            if torch.isnan(loss):
                raise RuntimeError("Nan encountered in the loss.")          # :283-284
        gradient_scaler.scale(loss).backward()                              # :287
        if (batch_id + 1) % accum_freq == 0:                                # :289
            if max_norm is not None:                                        # :290-295
                gradient_scaler.unscale_(optimizer)
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
            gradient_scaler.step(optimizer=optimizer)                       # :303
            gradient_scaler.update()                                        # :305
            optimizer.zero_grad(set_to_none=set_to_none)                    # :307
            iters += 1                                                      # :309
            if model_ema is not None:
                model_ema.update_parameters(model)                          # :311-312
        losses.append(float(loss.detach()))
    return iters - start_iteration, losses
