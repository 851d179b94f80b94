"""CPU: host-side mirror of the loss factory (no compute calls): LossCfg validation, create_loss weight arithmetic for
every scaling table of loss_functions.toml, error behaviour of the C ABI constructor."""
import pytest

from oracle import loss as ol


def test_losscfg_validation_matches_reference():
    from xmm_superres_denoise.config.config import ConfigError, LossCfg
    LossCfg(l1=0.0, poisson=0.0, psnr=0.5, ssim=0.0, ms_ssim=0.5)
    with pytest.raises(ConfigError):
        LossCfg(l1=0.0, poisson=0.0, psnr=0.0, ssim=0.0, ms_ssim=0.0)      # sum must be > 0
    with pytest.raises(ConfigError):
        LossCfg(l1=0.6, poisson=0.0, psnr=0.5, ssim=0.0, ms_ssim=0.0)      # and <= 1
    with pytest.raises(Exception):
        LossCfg(l1=-0.1, poisson=0.0, psnr=0.5, ssim=0.0, ms_ssim=0.5)


@pytest.mark.parametrize("scaling", ["linear", "sqrt", "asinh", "log"])
def test_create_loss_weights(scaling):
    from xmm_superres_denoise.utils import create_loss, load_loss_config
    sc, cfg = load_loss_config(scaling)
    loss = create_loss(sc, cfg)
    w, corr = ol.effective_weights(cfg.model_dump(), sc)
    assert {k: v for k, v in loss.weights.items() if v != 0.0} == pytest.approx(w)
    assert loss.correction == pytest.approx(corr)
    assert "psnr" in repr(loss) and "ms_ssim" in repr(loss)
    # no scaling table: plain percentages, no correction
    sc2, cfg2 = load_loss_config(scaling, use_scaling=False, l1=1.0, psnr=0.0, ms_ssim=0.0)
    plain = create_loss(sc2, cfg2)
    assert sc2 is None and plain.weights["l1"] == 1.0 and plain.correction == 0.0


def test_constructor_errors():
    from xmm_superres_denoise.engine._lib import XsdError
    from xmm_superres_denoise.utils import Loss
    with pytest.raises(XsdError):
        Loss({"l1": 0.0})
    with pytest.raises(XsdError):
        Loss({"vgg": 1.0})
    with pytest.raises(XsdError):
        Loss({"ssim": 1.0}, sigma=0.0)
