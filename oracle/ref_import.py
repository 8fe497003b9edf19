"""ref_import.py -- TEST INFRASTRUCTURE, build-container only.

Imports the real reference package from /root/reference (read-only, never copied) so that ``gen_golden.py`` can
produce reference outputs.  /root/reference does not exist on the GPU box: nothing in tests/, smoke() or bench.py may
import this module at run time.

The reference needs three modules that are not installed here: ``torchsearchsorted`` (third-party, un-pinned git
dependency -- requirements.txt:9; its searchsorted(a, v, side="right") is numpy's side='right', i.e.
torch.searchsorted(a, v, right=True)), and ``cv2`` / ``imageio`` (only touched by the dataset loaders).
"""
import contextlib
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nerf"))


def import_reference():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if "torchsearchsorted" not in sys.modules:
        ts = types.ModuleType("torchsearchsorted")
        ts.searchsorted = lambda a, v, side="left": torch.searchsorted(a, v, right=(side == "right"))
        sys.modules["torchsearchsorted"] = ts
    for name in ("cv2", "imageio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import nerf  # noqa: E402
    return nerf


def make_reference_model(nerf, cfg):
    """FlexibleNeRFModel with the missing `linear_layers` attribute supplied WITHOUT registering a sub-module
    (models.py:243 reads it): its length must equal num_layers."""
    m = nerf.models.FlexibleNeRFModel(
        num_layers=cfg["num_layers"], hidden_size=cfg["hidden_size"], skip_connect_every=cfg["skip_connect_every"],
        num_encoding_fn_xyz=cfg["num_encoding_fn_xyz"], num_encoding_fn_dir=cfg["num_encoding_fn_dir"],
        include_input_xyz=cfg.get("include_input_xyz", True), include_input_dir=cfg.get("include_input_dir", True),
        use_viewdirs=cfg.get("use_viewdirs", True))
    m.__dict__["linear_layers"] = [m.layer1] + list(m.layers_xyz)
    return m


@contextlib.contextmanager
def injected_randoms(draws):
    """Replace torch.rand / torch.randn by a replay of `draws` (a list of tensors, consumed in call order), so the
    reference consumes the same numbers the HIP path is given."""
    queue = list(draws)
    real_rand, real_randn = torch.rand, torch.randn

    def pop(shape):
        t = queue.pop(0)
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (list, tuple, torch.Size)) else tuple(shape)
        assert tuple(t.shape) == shape, "draw shape %s != requested %s" % (tuple(t.shape), shape)
        return t.clone()

    def fake_rand(*shape, **kw):
        return pop(shape)

    def fake_randn(*shape, **kw):
        return pop(shape)

    torch.rand, torch.randn = fake_rand, fake_randn
    try:
        yield
    finally:
        torch.rand, torch.randn = real_rand, real_randn
    assert not queue, "%d injected draws were not consumed" % len(queue)
