"""A minimal attribute-access options tree.  The hot path only ever *reads* attributes of the reference's CfgNode
(options.nerf.<mode>.num_coarse, options.dataset.near, ... -- SURVEY 5.6); the reference's own CfgNode (or any object
with the same attributes) can be passed instead."""


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def make_options(num_coarse=64, num_fine=128, perturb=True, lindisp=False, white_background=False,
                 radiance_field_noise_std=0.2, chunksize=131072, num_random_rays=4096, use_viewdirs=True, no_ndc=True,
                 near=2.0, far=6.0):
    """Options with the keys the hot path reads (config/lego.yml layout): nerf.use_viewdirs,
    nerf.{train,validation}.*, dataset.{no_ndc,near,far}."""
    blk = dict(num_random_rays=num_random_rays, chunksize=chunksize, perturb=perturb, num_coarse=num_coarse,
               num_fine=num_fine, white_background=white_background, radiance_field_noise_std=radiance_field_noise_std,
               lindisp=lindisp)
    return AttrDict(nerf=dict(use_viewdirs=use_viewdirs, train=dict(blk), validation=dict(blk)),
                    dataset=dict(no_ndc=no_ndc, near=near, far=far))
