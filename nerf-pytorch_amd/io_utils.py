"""Checkpoint and cached-dataset formats of the reference, both directions.

* ``.ckpt`` -- the dict ``train_nerf.py:373-388`` saves and ``train_nerf.py:156-163`` / ``eval_nerf.py:128-143`` load:
  ``iter, model_coarse_state_dict, model_fine_state_dict (None without a fine net), optimizer_state_dict, loss, psnr``
  (+ optional ``height, width, focal_length`` that eval_nerf.py honours).  The optimizer entry is a
  ``torch.optim.Adam.state_dict()`` over ``list(model_coarse.parameters()) + list(model_fine.parameters())``
  (train_nerf.py:138-143); TrainEngine keeps the same moments in two flat buffers, converted here.  Both layouts of
  that dict are read: torch 1.x (state keyed by ``id(param)``, integer ``step``) as in the reference's
  ``pretrained/*`` files, and the current one (state keyed by position, tensor ``step``).
* ``.data`` -- the per-image dicts ``cache_dataset.py:104-135`` writes: train ``{height, width, focal_length,
  ray_bundle (2, ., 3), target}``, val ``{height, width, focal_length, ray_origins, ray_directions, target}``.
"""
import torch


def _engine_models(engine):
    return [engine.mc] + ([engine.mf] if engine.mf is not None else [])


def _engine_params(engine):
    out = []
    for m in _engine_models(engine):
        out += list(m.parameters())
    return out


def engine_optimizer_state_dict(engine):
    """TrainEngine's Adam state as the ``torch.optim.Adam.state_dict()`` the reference's script would hold."""
    params = _engine_params(engine)
    b1, b2 = engine.betas
    opt = torch.optim.Adam(params, lr=engine.lr, betas=(b1, b2), eps=engine.eps)
    sd = opt.state_dict()
    off = 0
    for i, p in enumerate(params):
        n = p.numel()
        if engine.step_count > 0:
            sd["state"][i] = {"step": torch.tensor(float(engine.step_count)),
                              "exp_avg": engine.exp_avg[off:off + n].view(p.shape).clone(),
                              "exp_avg_sq": engine.exp_avg_sq[off:off + n].view(p.shape).clone()}
        off += n
    if off != engine.exp_avg.numel():
        raise RuntimeError("parameter order does not cover the engine's flat buffers")
    return sd


def load_engine_optimizer_state(engine, state_dict):
    """Inverse of engine_optimizer_state_dict; also reads the torch 1.x layout of the reference's own checkpoints."""
    params = _engine_params(engine)
    group = state_dict["param_groups"][0]
    keys = list(group["params"])
    if len(keys) != len(params):
        raise ValueError("optimizer state holds %d parameters, the models have %d" % (len(keys), len(params)))
    state = state_dict["state"]
    steps = set()
    off = 0
    engine.exp_avg.zero_()
    engine.exp_avg_sq.zero_()
    for key, p in zip(keys, params):
        n = p.numel()
        st = state.get(key)
        if st is not None:
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state shape %s does not match parameter %s" % (tuple(st["exp_avg"].shape),
                                                                                          tuple(p.shape)))
            engine.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            engine.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        off += n
    if len(steps) > 1:
        raise ValueError("per-parameter Adam steps differ: %s" % sorted(steps))
    engine.step_count = steps.pop() if steps else 0
    engine.lr = float(group.get("lr", engine.lr))
    engine.betas = tuple(group.get("betas", engine.betas))
    engine.eps = float(group.get("eps", engine.eps))


def save_checkpoint(path, iteration, model_coarse, model_fine, optimizer_state_dict, loss, psnr, **extra):
    """train_nerf.py:373-388.  `extra`: e.g. height=, width=, focal_length= (read by eval_nerf.py:138-143)."""
    ck = {"iter": iteration,
          "model_coarse_state_dict": model_coarse.state_dict(),
          "model_fine_state_dict": None if not model_fine else model_fine.state_dict(),
          "optimizer_state_dict": optimizer_state_dict,
          "loss": loss,
          "psnr": psnr}
    ck.update(extra)
    torch.save(ck, path)
    return ck


def load_checkpoint(path, model_coarse, model_fine=None, engine=None, map_location=None):
    """train_nerf.py:156-163 / eval_nerf.py:128-143: restores the nets (and, given a TrainEngine, the Adam state and
    the re-packed weights).  Returns the checkpoint dict (``start_iter = ck["iter"]``)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model_coarse.load_state_dict(ck["model_coarse_state_dict"])
    if ck["model_fine_state_dict"] and model_fine is not None:
        model_fine.load_state_dict(ck["model_fine_state_dict"])
    if engine is not None:
        if ck.get("optimizer_state_dict"):
            load_engine_optimizer_state(engine, ck["optimizer_state_dict"])
        engine.repack()
    return ck


def save_cached_example(path, height, width, focal_length, target, ray_bundle=None, ray_origins=None,
                        ray_directions=None):
    """cache_dataset.py:104-135: train files carry the stacked bundle, validation files the two maps."""
    d = {"height": height, "width": width, "focal_length": focal_length}
    if ray_bundle is not None:
        d["ray_bundle"] = ray_bundle.detach().cpu()
    else:
        d["ray_origins"] = ray_origins.detach().cpu()
        d["ray_directions"] = ray_directions.detach().cpu()
    d["target"] = target.detach().cpu()
    torch.save(d, path)
    return d


def load_cached_example(path, device=None):
    """train_nerf.py:176-183, :287-296: the dict, tensors moved to `device`."""
    d = torch.load(path, map_location="cpu", weights_only=False)
    if device is not None:
        d = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    return d
