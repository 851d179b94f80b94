"""MI355X-native drop-in for the conv hot path of SamSweere/xmm-superres-denoise.

Mirrors the reference package layout for the pieces on the hot path:
  xmm_superres_denoise.models      -> Model, GeneratorRRDB_DN, GeneratorRRDB_SR   (reference models/__init__.py:1-2)
  xmm_superres_denoise.transforms  -> Crop, ImageUpsample, Normalize               (reference transforms/__init__.py:1-3)
  xmm_superres_denoise.config      -> RrdbCfg, ModelCfg, OptimizerCfg, BaseModels  (reference config/config.py:164-203)
  xmm_superres_denoise.data.tools  -> reshape_img_to_res, load_fits                (reference data/tools.py:79-126)
Compute runs in libxsd_hip.so (hand-written HIP for gfx950) through the C ABI in include/xsd.h; there is no CPU
fallback: calling a model without the library or with CPU tensors raises.
"""
__version__ = "0.1.0"
