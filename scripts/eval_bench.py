"""GPU-box measurement for BASELINE config 5: inference-only render of full 800x800 views (640,000 rays per image,
64 coarse + 128 fine, 8x256 nets) through run_one_iter_of_nerf(mode="validation") -- forward kernels only, no stash."""
import json, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_pytorch_amd as N
sys.path.insert(0, ROOT)
from bench import pose_spherical, MODEL
dev = torch.device("cuda", 0)
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 800
focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
torch.manual_seed(42)
mc, mf = N.FlexibleNeRFModel(**MODEL).to(dev), N.FlexibleNeRFModel(**MODEL).to(dev)
ex, ed = N.get_embedding_function(10, True, True), N.get_embedding_function(4, True, True)
opts = N.make_options(64, 128, perturb=False, radiance_field_noise_std=0.0, chunksize=131072)
from nerf_pytorch_amd.eval_utils import ImageWriter
import tempfile
outdir = tempfile.mkdtemp(prefix="nerfhip_eval_")
writer = ImageWriter(workers=4)
times = []
t_all = None
with torch.no_grad():
    for i, th in enumerate((-180.0, -90.0, 0.0, 90.0)):
        pose = pose_spherical(th, -30.0, 4.0).to(dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ro, rd = N.get_ray_bundle(H, W, focal, pose[:3, :4])
        out = N.run_one_iter_of_nerf(H, W, focal, mc, mf, ro, rd, opts, mode="validation", encode_position_fn=ex,
                                     encode_direction_fn=ed)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
    # the same loop with eval_nerf.py's output stage attached (8-bit casts on the device, async D2H, PNG encoding on
    # worker threads): wall time per image including every file on disk
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, th in enumerate((-135.0, -45.0, 45.0, 135.0)):
        pose = pose_spherical(th, -30.0, 4.0).to(dev)
        ro, rd = N.get_ray_bundle(H, W, focal, pose[:3, :4])
        out = N.run_one_iter_of_nerf(H, W, focal, mc, mf, ro, rd, opts, mode="validation", encode_position_fn=ex,
                                     encode_direction_fn=ed)
        writer.submit(os.path.join(outdir, "%04d.png" % i), out[3][..., :3])
        writer.submit(os.path.join(outdir, "disparity", "%04d.png" % i), out[4], disparity=True)
    files = writer.close()
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 4
best = min(times)
flops = 2 * 593408 * (64 + 192) * H * W
print(json.dumps(dict(what="eval render", H=H, W=W, rays=H * W, s_per_image=best, rays_per_s=H * W / best,
                      tflops=flops / best / 1e12, frac_fp32_mfma_peak=flops / best / 1e12 / 157.3,
                      finite=bool(torch.isfinite(out[3]).all()), shape=list(out[3].shape),
                      s_per_image_with_png_output=t_all, files_written=len(files),
                      png_bytes=sum(os.path.getsize(f) for f in files))))
