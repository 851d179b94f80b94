"""`Model`: the reference's LightningModule (models/model.py:13-247) restated without Lightning for the RRDB paths:
constructor signature, `configure_model` factory (model.py:153-186), `forward` = clamp(generator(x), 0, 1)
(model.py:48-49 -- the second clamp is fused in the engine's output kernel and is idempotent), `_on_step` / `_on_epoch_end` (model.py:72-150, returning what the reference logs) and `configure_optimizers`
(model.py:239-247).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from xmm_superres_denoise.config.config import BaseModels, ModelCfg


def _mean_over_ranks(values: dict) -> dict:
    """`sync_dist=True` semantics of the reference's logging (models/model.py:118,150) for a metric collection that cannot
    reduce its own states: every rank contributes every key, in sorted order, so the collectives match."""
    import torch.distributed as dist
    from xmm_superres_denoise.parallel import collectives_on
    if not collectives_on() or not values:
        return values
    keys = sorted(values)
    t = torch.stack([torch.as_tensor(values[k]).detach().double().reshape(()) for k in keys])
    if dist.get_backend() == "nccl":
        t = t.cuda()
    else:
        t = t.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = (t / dist.get_world_size()).float()
    return {k: t[i] for i, k in enumerate(keys)}


class Model(nn.Module):
    def __init__(self, config: ModelCfg, lr_shape: Tuple[int, int], hr_shape: Tuple[int, int], loss=None,
                 metrics=None, extended_metrics=None, in_metrics=None, in_extended_metrics=None):
        super().__init__()
        self.config = config
        self.metrics = metrics
        self.ext_metrics = extended_metrics
        self.in_metrics = in_metrics
        self.in_ext_metrics = in_extended_metrics
        self.loss = loss
        self.model: Optional[nn.Module] = None
        self.hr_shape = hr_shape
        self.lr_shape = lr_shape

    def forward(self, x) -> torch.Tensor:
        if self.model is None:
            self.configure_model()
        # the generator already returns clamp(clamp(.)) == clamp(.)
        return self.model(x)

    def configure_model(self) -> None:
        if self.model is not None:
            return
        from xmm_superres_denoise.models import GeneratorRRDB_DN, GeneratorRRDB_SR
        name = BaseModels(self.config.name)
        if name is BaseModels.ESR_GEN:
            up_scale = self.hr_shape[0] / self.lr_shape[0]
            if up_scale % 2 != 0:
                raise ValueError(f"Upscaling is not a multiple of two but {up_scale}, "
                                 f"based on in_dims {self.lr_shape} and out_dims {self.hr_shape}")
            self.model = GeneratorRRDB_SR(in_channels=self.config.model.in_channels,
                                          out_channels=self.config.model.out_channels,
                                          num_filters=self.config.model.filters,
                                          num_res_blocks=self.config.model.residual_blocks,
                                          num_upsample=int(up_scale / 2),
                                          memory_efficient=self.config.memory_efficient)
        elif name is BaseModels.RRDB_DENOISE:
            self.model = GeneratorRRDB_DN(in_channels=self.config.model.in_channels,
                                          out_channels=self.config.model.out_channels,
                                          num_filters=self.config.model.filters,
                                          num_res_blocks=self.config.model.residual_blocks,
                                          memory_efficient=self.config.memory_efficient)
        else:
            raise NotImplementedError(f"{name}: only the RRDB generators are on the MI355X hot path (SURVEY.md section 8)")

    def training_step(self, batch, batch_idx=0):
        return self._on_step(batch, "train")

    def on_validation_start(self) -> None:
        self._on_epoch_end("train")

    def validation_step(self, batch, batch_idx=0):
        self._on_step(batch, "val")

    def on_validation_epoch_end(self) -> dict:
        return self._on_epoch_end("val")

    def test_step(self, batch, batch_idx=0):
        self._on_step(batch, "test")

    def on_test_epoch_end(self) -> dict:
        return self._on_epoch_end("test")

    def predict_step(self, batch, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        batch["preds"] = self(batch["lr"])

    def _on_step(self, batch, stage):
        """reference models/model.py:72-107: train -> loss(preds, target); val/test -> update the epoch states of the
        loss and of the metric collections (input metrics on the nearest-upsampled low-resolution image)."""
        lr_img, hr_img = batch
        preds = self(lr_img)
        target = hr_img if hr_img is not None else preds
        if stage == "train":
            return self.loss(preds, target)
        with torch.no_grad():
            self.loss.update(preds=preds, target=target)
            if self.in_metrics is not None or self.ext_metrics is not None:
                scale_factor = target.shape[2] / lr_img.shape[2]
                if scale_factor != 1.0:
                    from xmm_superres_denoise.transforms import ImageUpsample
                    lr_img = ImageUpsample(scale_factor=int(scale_factor))(lr_img)
            if self.metrics is not None:
                self.metrics.update(preds=preds, target=target)
            if self.in_metrics is not None:
                self.in_metrics.update(preds=lr_img, target=target)
            if self.ext_metrics is not None:
                self.ext_metrics.update(preds=preds, target=target)
            if self.in_ext_metrics is not None:
                self.in_ext_metrics.update(preds=lr_img, target=target)
        return None

    def _on_epoch_end(self, stage):
        """reference models/model.py:109-150 without the Lightning logger: returns what it would log.  With several ranks
        the metric STATES are reduced first (sum / min / max, what torchmetrics' dist_reduce_fx does at compute()), so every
        rank returns the global epoch value."""
        if stage == "train":
            if self.loss is not None:
                self.loss.reset()
            return {}
        self.loss.sync()
        logged = {f"{stage}/loss": self.loss.compute()}
        self.loss.reset()
        for name in ("metrics", "ext_metrics", "in_metrics", "in_ext_metrics"):
            coll = getattr(self, name)
            if coll is not None:
                if hasattr(coll, "sync"):
                    coll.sync()
                    logged.update(coll.compute())
                else:       # a collection without state reduction: fall back to the mean over the ranks of its values
                    logged.update(_mean_over_ranks(coll.compute()))
                coll.reset()
                if name.startswith("in_"):
                    setattr(self, name, None)      # input metrics are only needed once (reference :135-142)
        self.logged = logged
        return logged

    def configure_optimizers(self):
        if self.model is None:
            self.configure_model()
        return torch.optim.Adam(self.model.parameters(), lr=self.config.optimizer.learning_rate,
                                betas=tuple(self.config.optimizer.betas))
