"""Training driver: the engine-side counterpart of the reference's `train.py fit` (train.py:19-171) for the two RRDB
models, without Lightning.  One process per GPU (launch with torch.distributed.run for N > 1); data parallelism and the
optimizer are `parallel.DataParallelTrainer` (flat-buffer RCCL all-reduce overlapped with backward + fused Adam).

The data feed here is the reference's `BoringDataset` analogue (random tensors of the configured shapes,
data/dataset.py:52-74): file discovery / FITS matching (`XmmDataset.__init__`) is host I/O outside the hot path; real
samples are composed on the GPU with `engine.compose_input` (img + agn + background, mask, pad, normalize).

Checkpoints use the reference's Lightning layout: {"state_dict": {"model.<key>": tensor}} with the reference key names,
so `Model.load_from_checkpoint`-style consumers (utils/run_inference_on_file.py:28-35) can read them.
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.distributed as dist

from xmm_superres_denoise.config.config import model_cfg
from xmm_superres_denoise.models import Model
from xmm_superres_denoise.parallel import DataParallelTrainer


def save_checkpoint(path: str, model: Model, trainer: DataParallelTrainer, epoch: int) -> None:
    sd = {"model." + k: v.detach().cpu() for k, v in model.model.state_dict().items()}
    torch.save({"state_dict": sd, "epoch": epoch, "global_step": trainer.step_count,
                "adam": {"m": trainer.m.cpu(), "v": trainer.v.cpu(), "step": trainer.step_count}}, path)


def load_checkpoint(path: str, model: Model, trainer: DataParallelTrainer | None = None) -> dict:
    """Reads a checkpoint with the loader that executes nothing from the file (`weights_only=True`: tensors, numbers, strings,
    dicts / lists of them).  What save_checkpoint writes is exactly that; a Lightning `.ckpt` carrying pickled objects
    (callback states, hyper-parameter namespaces) is refused rather than unpickled: the error raised here names what is accepted
    and the one-liner that reduces such a file to {"state_dict": ...} on a machine the user trusts."""
    import pickle
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:      # a Lightning .ckpt with objects under hyper_parameters / callbacks
        raise RuntimeError(
            f"{path}: the safe loader (weights_only=True) refused this checkpoint -- it holds pickled objects besides tensors and plain "
            "containers (a genuine Lightning .ckpt carries them under `hyper_parameters` / `callbacks`).  Accepted: a file whose "
            "`state_dict` maps the reference's key names (with or without Lightning's `model.` prefix) to tensors, optionally with this "
            "driver's `adam` block.  Reduce it on a machine you trust: torch.save({'state_dict': torch.load(path, weights_only=False)"
            f"['state_dict']}}, out).  ({e})") from e
    if not isinstance(ck, dict) or "state_dict" not in ck:
        raise RuntimeError(f"{path}: no `state_dict` in this checkpoint (keys: {sorted(ck) if isinstance(ck, dict) else type(ck).__name__})")
    raw = ck["state_dict"]
    # Lightning prefixes the generator's keys with the attribute name (`model.`, models/model.py:157-186); a bare state_dict is taken as is
    sd = {k[len("model."):]: v for k, v in raw.items() if k.startswith("model.")} or dict(raw)
    if model.model is None:
        model.configure_model()
    model.model.load_state_dict(sd)
    if trainer is not None and "adam" in ck:
        trainer.m.copy_(ck["adam"]["m"]); trainer.v.copy_(ck["adam"]["v"]); trainer.step_count = int(ck["adam"]["step"])
    return ck


def fit(name: str = "rrdb_denoise", lr_res: int = 416, batch_size: int = 4, steps: int = 10, device: str | None = None,
        checkpoint: str | None = None, seed: int = 0, math: str | None = None, log_every: int = 1,
        loss: str = "l1", scaling: str = "linear", val_batches: int = 0):
    """loss: "l1" (BASELINE configs[2]) or "paper" = the reference's shipped default, 0.5 psnr + 0.5 ms_ssim with the
    scaling table of the dataset's stretch mode (`scaling`; res/configs/loss_functions.toml, train.py:46-63)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU over RCCL ("nccl"); XSD_DIST_BACKEND=gloo rehearses N ranks on fewer GPUs (ranks then share devices),
    # exactly as bench.py does
    backend = os.environ.get("XSD_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise RuntimeError("train.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    if backend == "nccl" and world > ndev:
        raise RuntimeError(f"{world} ranks but {ndev} GPU(s): RCCL needs one GPU per rank (XSD_DIST_BACKEND=gloo rehearses the "
                           "multi-rank path on fewer GPUs)")
    dev = torch.device(device or f"cuda:{local_rank if backend == 'nccl' else local_rank % ndev}")
    torch.cuda.set_device(dev)
    force_dp = os.environ.get("XSD_FORCE_DP", "0") == "1"      # one rank, collectives on: the RCCL path on a single GPU (parallel.collectives_on)
    if (world > 1 or force_dp) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cfg = model_cfg(name, batch_size=batch_size)
    hr_res = lr_res * (2 if name == "esr_gen" else 1)
    torch.manual_seed(seed)
    from xmm_superres_denoise.utils import Loss, create_loss, load_loss_config
    loss_fn = None
    if loss != "l1":
        loss_fn = create_loss(*load_loss_config(scaling))
    metrics = None
    if val_batches > 0:   # train.py:86-104 of the reference: metric collections over the scaling normalizers
        from xmm_superres_denoise.metrics import get_metrics
        from xmm_superres_denoise.transforms import Normalize
        norm = Normalize(lr_max=0.0022336, hr_max=0.0022336 if name == "rrdb_denoise" else 0.0005584, stretch_mode=scaling)
        metrics = get_metrics(norm, [Normalize(norm.lr_max.item(), norm.hr_max.item(), "linear")], "val")
    model = Model(cfg, (lr_res, lr_res), (hr_res, hr_res), loss=loss_fn if loss_fn is not None else Loss({"l1": 1.0}),
                  metrics=metrics, extended_metrics=None,
                  in_metrics=None, in_extended_metrics=None)
    model.configure_model()
    model.to(dev)
    if math:
        model.model.set_math(math)
    trainer = DataParallelTrainer(model.model, lr=cfg.optimizer.learning_rate, betas=cfg.optimizer.betas, loss=loss_fn)
    per_rank = batch_size // world if batch_size % world == 0 else batch_size
    g = torch.Generator().manual_seed(seed + 1 + rank)
    losses = []
    for it in range(steps):
        lr_img = torch.rand((per_rank, 1, lr_res, lr_res), generator=g).to(dev)   # BoringDataset analogue
        hr_img = torch.rand((per_rank, 1, hr_res, hr_res), generator=g).to(dev)
        loss = trainer.global_loss(trainer.train_step(lr_img, hr_img))
        losses.append(float(loss))
        if rank == 0 and log_every and it % log_every == 0:
            print(f"step {it}: train/loss {losses[-1]:.6f}", flush=True)
    if val_batches > 0:       # one validation epoch (reference model.py:53-60); metric states are reduced over the ranks
        model.on_validation_start()                                    # inside on_validation_epoch_end (sum / min / max)
        for _ in range(val_batches):
            lr_img = torch.rand((per_rank, 1, lr_res, lr_res), generator=g).to(dev)
            hr_img = torch.rand((per_rank, 1, hr_res, hr_res), generator=g).to(dev)
            model.validation_step((lr_img, hr_img))
        logged = model.on_validation_epoch_end()
        if rank == 0 and log_every:
            print("validation: " + ", ".join(f"{k} {float(v):.6f}" for k, v in sorted(logged.items())), flush=True)
    if checkpoint and rank == 0:
        save_checkpoint(checkpoint, model, trainer, epoch=0)
    return model, trainer, losses


def main():
    ap = argparse.ArgumentParser(description="fit an RRDB generator on synthetic tiles with the MI355X engine")
    ap.add_argument("routine", choices=["fit"])
    ap.add_argument("--model", default="rrdb_denoise", choices=["rrdb_denoise", "esr_gen"])
    ap.add_argument("--lr-res", type=int, default=416)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--math", default=None, choices=[None, "fp32", "bf16x6", "f16x3"])
    ap.add_argument("--loss", default="l1", choices=["l1", "paper"], help="paper = 0.5 psnr + 0.5 ms_ssim (loss_functions.toml)")
    ap.add_argument("--scaling", default="linear", choices=["linear", "sqrt", "asinh", "log"])
    ap.add_argument("--val-batches", type=int, default=0, help="validation batches after training (loss + metric set)")
    a = ap.parse_args()
    fit(a.model, a.lr_res, a.batch_size, a.steps, checkpoint=a.checkpoint, math=a.math, loss=a.loss, scaling=a.scaling, val_batches=a.val_batches)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
