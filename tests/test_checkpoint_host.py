"""Checkpoint loading executes nothing from the file (train.load_checkpoint; reference: Lightning's ModelCheckpoint layout,
train.py:133-139, read back by Model.load_from_checkpoint, utils/run_inference_on_file.py:28-35).  CPU only."""
import os
import pickle

import pytest
import torch


class _Payload:            # stands for anything a third-party .ckpt may carry besides tensors (callback state, namespaces, ...)
    def __init__(self):
        self.note = "would run on unpickling"


def _tiny_model():
    from xmm_superres_denoise.config.config import model_cfg
    from xmm_superres_denoise.models import Model
    m = Model(model_cfg("rrdb_denoise", batch_size=1, residual_blocks=1), (32, 32), (32, 32), None, None, None, None, None)
    m.configure_model()
    return m


def test_checkpoint_of_tensors_and_numbers_loads_with_the_safe_loader(tmp_path):
    from xmm_superres_denoise.train import load_checkpoint
    torch.manual_seed(3)
    a, b = _tiny_model(), _tiny_model()
    sd = {"model." + k: v.detach().clone() for k, v in a.model.state_dict().items()}
    n = sum(v.numel() for v in sd.values())
    p = os.path.join(tmp_path, "ok.ckpt")
    torch.save({"state_dict": sd, "epoch": 3, "global_step": 17, "adam": {"m": torch.zeros(n), "v": torch.ones(n), "step": 17}}, p)
    ck = load_checkpoint(p, b)
    assert ck["epoch"] == 3 and ck["adam"]["step"] == 17
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[k]), k


def test_checkpoint_with_a_non_tensor_payload_is_refused(tmp_path):
    from xmm_superres_denoise.train import load_checkpoint
    m = _tiny_model()
    before = {k: v.detach().clone() for k, v in m.model.state_dict().items()}
    sd = {"model." + k: torch.zeros_like(v) for k, v in before.items()}
    p = os.path.join(tmp_path, "third_party.ckpt")
    torch.save({"state_dict": sd, "callbacks": _Payload()}, p)
    # refused, with a message that says what is accepted and how to reduce the file (round 6; the cause is torch's UnpicklingError)
    with pytest.raises(RuntimeError, match="weights_only=True") as ei:
        load_checkpoint(p, m)
    assert isinstance(ei.value.__cause__, pickle.UnpicklingError) and "state_dict" in str(ei.value) and "hyper_parameters" in str(ei.value)
    for k, v in before.items():      # refused before anything was applied
        assert torch.equal(v, m.model.state_dict()[k]), k


def test_bare_state_dict_without_the_lightning_prefix_loads(tmp_path):
    from xmm_superres_denoise.train import load_checkpoint
    torch.manual_seed(4)
    a, b = _tiny_model(), _tiny_model()
    p = os.path.join(tmp_path, "bare.ckpt")
    torch.save({"state_dict": {k: v.detach().clone() for k, v in a.model.state_dict().items()}}, p)
    load_checkpoint(p, b)
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[k]), k
    q = os.path.join(tmp_path, "nothing.ckpt")
    torch.save({"weights": 1}, q)
    with pytest.raises(RuntimeError, match="no `state_dict`"):
        load_checkpoint(q, b)
