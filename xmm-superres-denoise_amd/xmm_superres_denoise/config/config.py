"""The slice of the reference's pydantic config that defines the constructor-API contract of the hot path
(reference config/config.py:36-42,164-203).  Python 3.10 here: enum.StrEnum restated as (str, Enum)."""
from enum import Enum
from typing import Literal, Tuple

from pydantic import BaseModel, Field, NonNegativeFloat, PositiveInt, model_validator


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


class ConfigError(Exception):
    """reference config/config.py:19-21"""

    def __init__(self, message: str = ""):
        super().__init__(message)


class LossCfg(BaseModel):
    """reference config/config.py:222-237: relative percentages of the loss terms, 0 < sum <= 1"""
    l1: float = Field(ge=0, le=1)
    poisson: float = Field(ge=0, le=1)
    psnr: float = Field(ge=0, le=1)
    ssim: float = Field(ge=0, le=1)
    ms_ssim: float = Field(ge=0, le=1)

    @model_validator(mode="after")
    def check_sum(self):
        p_sum = self.l1 + self.poisson + self.psnr + self.ssim + self.ms_ssim
        if 0 < p_sum <= 1:
            return self
        raise ConfigError(f"Sum of relative percentages has to be between 0 and 1, got {p_sum}!")
