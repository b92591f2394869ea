"""One rank of tests/test_dropin_reference_loop.py::test_reference_loop_under_accelerate_ddp (launched by torch.distributed.run, CPU, gloo):
the reference's pipeline/train/instruction_following.py is imported through shim/, the model is otter_amd's, `accelerate` wraps it in
DistributedDataParallel, and the reference's own train_one_epoch runs two optimizer steps on this rank's batches.  Test infrastructure: the
fusion modules' arithmetic comes from the numpy oracle (tests/_cpu_backend.py) because the product has no CPU path."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import test_dropin_reference_loop as T  # noqa: E402


def main(outdir):
    IF, _restore = T.install_reference_script()
    from accelerate import Accelerator

    from tests._cpu_backend import oracle_backend

    tu = sys.modules["pipeline.train.train_utils"]
    accelerator = Accelerator(gradient_accumulation_steps=1, mixed_precision="no", cpu=True)
    rank, world = accelerator.process_index, accelerator.num_processes
    torch.set_num_threads(2)
    model = T._build()
    batches = T._batches(model, 2, seed0=11 + 100 * rank)
    tokenizer = model.text_tokenizer
    args = types.SimpleNamespace(model_name="otter", total_training_steps=2, gradient_accumulation_steps=1, rank=rank, world_size=world, batch_size=2,
                                 remove_answer_token=False, remove_eos_token=False, mask_lm_head=False, distributed_type=str(accelerator.distributed_type),
                                 report_to_wandb=False, save_steps_interval=-1, logging_steps=1, num_epochs=1, external_save_dir=outdir, save_hf_model=False)
    optimizer = torch.optim.AdamW(tu.get_grouped_params(model, wd=0.1), lr=1e-3)
    sched = IF.get_constant_schedule_with_warmup(optimizer, num_warmup_steps=0)
    losses = []
    orig_forward = model.forward

    def recording_forward(*a, **k):
        out = orig_forward(*a, **k)
        losses.append(float(out[0].detach()))
        return out

    model.forward = recording_forward
    # what the reference's main() does before the loop (instruction_following.py:470-494)
    ddp_model, optimizer, sched = accelerator.prepare(model, optimizer, sched)
    with oracle_backend():
        IF.train_one_epoch(args, ddp_model, 0, [T._Loader(batches)], tokenizer, optimizer, sched, accelerator.device, accelerator, sys.modules["wandb"])
    accelerator.wait_for_everyone()
    unwrapped = accelerator.unwrap_model(ddp_model)
    assert unwrapped is model
    if rank == 0:
        torch.save({"world": world, "wrapped": type(ddp_model).__name__, "losses": losses,
                    "weights": {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}}, os.path.join(outdir, "ddp_rank0.pt"))
    accelerator.wait_for_everyone()


if __name__ == "__main__":
    main(sys.argv[1])
