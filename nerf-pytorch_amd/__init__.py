"""nerf-pytorch_amd -- the MI355X-native NeRF render + training hot path behind the krrish94/nerf-pytorch API.

Import as ``nerf_pytorch_amd`` (the alias package next to this directory).  Mirrors the reference's star-export
surface for the hot path (nerf/__init__.py:1-7): nerf_helpers, volume_rendering_utils, train_utils, models.
Next to the path: on-device training-ray selection (train_utils.select_training_rays), the 8-bit
output stage (eval_utils) and the reference's .ckpt / .data formats (io_utils).  Dataset loaders, the YAML config
tree and the CLI scripts of the reference are out of scope.
"""
from . import eval_utils, io_utils, models  # noqa: F401
from .cfg import AttrDict, make_options  # noqa: F401
from .engine import TrainEngine  # noqa: F401
from .models import FlexibleNeRFModel  # noqa: F401
from .nerf_helpers import (cumprod_exclusive, get_embedding_function, get_minibatches, get_ray_bundle,  # noqa: F401
                           get_rays_at_pixels, img2mse, meshgrid_xy, mse2psnr, ndc_rays, positional_encoding,
                           sample_pdf, sample_pdf_2, sample_pdf_with_indices)
from .eval_utils import ImageWriter, cast_to_disparity_image, cast_to_image, render_pose_rows  # noqa: F401
from .io_utils import load_cached_example, load_checkpoint, save_cached_example, save_checkpoint  # noqa: F401
from .train_utils import (pack_rays, predict_and_render_radiance, run_network, run_one_iter_of_nerf,  # noqa: F401
                          select_cached_training_rays, select_training_rays)
from .volume_rendering_utils import volume_render_radiance_field  # noqa: F401

__version__ = "0.1.0"
