"""The slice of the reference's pydantic config that defines the constructor-API contract of the hot path
(reference config/config.py:36-42,164-203).  Python 3.10 here: enum.StrEnum restated as (str, Enum)."""
from enum import Enum
from typing import Literal, Tuple

from pydantic import BaseModel, NonNegativeFloat, PositiveInt


class BaseModels(str, Enum):
    ESR_GEN = "esr_gen"
    RRDB_DENOISE = "rrdb_denoise"
    SWINFIR = "swinfir"
    DRCT = "drct"
    HAT = "hat"
    RESTORMER = "restormer"

    def __str__(self):
        return self.value


class OptimizerCfg(BaseModel):
    learning_rate: NonNegativeFloat
    betas: Tuple[NonNegativeFloat, NonNegativeFloat]


class RrdbCfg(BaseModel):
    base_model: Literal["esr_gen", "rrdb_denoise"]
    in_channels: PositiveInt
    out_channels: PositiveInt
    filters: PositiveInt
    residual_blocks: PositiveInt


class ModelCfg(BaseModel):
    name: BaseModels
    memory_efficient: bool
    batch_size: PositiveInt
    model: RrdbCfg  # reference: RrdbCfg | TransformerCfg | RestormerCfg (transformer zoo is off the hot path)
    optimizer: OptimizerCfg


# res/configs/models.toml:1-17 of the reference (the two shipped RRDB models)
MODELS_TOML = {
    "esr_gen": dict(base_model="esr_gen", in_channels=1, out_channels=1, filters=32, residual_blocks=4,
                    learning_rate=0.0001, betas=(0.9, 0.999)),
    "rrdb_denoise": dict(base_model="rrdb_denoise", in_channels=1, out_channels=1, filters=32, residual_blocks=4,
                         learning_rate=0.0001, betas=(0.9, 0.999)),
}


def model_cfg(name: str, batch_size: int = 1, memory_efficient: bool = False, **overrides) -> ModelCfg:
    """What train.py:35-44 of the reference does: merge the run config's model section with models.toml[name]
    and split the optimizer fields out."""
    d = dict(MODELS_TOML[name])
    d.update(overrides)
    opt = OptimizerCfg(learning_rate=d.pop("learning_rate"), betas=tuple(d.pop("betas")))
    return ModelCfg(name=BaseModels(name), memory_efficient=memory_efficient, batch_size=batch_size,
                    model=RrdbCfg(**d), optimizer=opt)
