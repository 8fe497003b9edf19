// nh_rays.h -- per-ray device arithmetic shared by the unit kernels (elementwise.hip) and the fused training-ray
// selection (select.hip), so that both produce identical bits.
#pragma once
#include "nh_device.h"

// get_ray_bundle (nerf/nerf_helpers.py:67-110)
// one pin-hole ray: pixel (row, col) of a height x width image
NH_DEVICE void nh_pinhole_ray(int height, int width, float focal, const float* __restrict__ c2w, int ld, int64_t row,
                              int64_t col, float* o, float* d) {
    float ii = (float)col;  // x
    float jj = (float)row;  // y
    float dx = (ii - (float)(width * 0.5)) / focal;
    float dy = -(jj - (float)(height * 0.5)) / focal;
    float dz = -1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = dx * c2w[c * ld + 0];
        v = v + dy * c2w[c * ld + 1];
        v = v + dz * c2w[c * ld + 2];
        d[c] = v;
        o[c] = c2w[c * ld + 3];
    }
}

// ndc_rays (nerf/nerf_helpers.py:170-197)
struct NhNdc {
    float near, cw, ch, two_near, neg_two_near;
};
NH_DEVICE void nh_ndc_ray(const NhNdc& k, float* o, float* d) {
    float ox = o[0], oy = o[1], oz = o[2];
    float dx = d[0], dy = d[1], dz = d[2];
    float t = -(k.near + oz) / dz;
    ox = ox + t * dx;
    oy = oy + t * dy;
    oz = oz + t * dz;
    o[0] = k.cw * ox / oz;
    o[1] = k.ch * oy / oz;
    o[2] = 1.0f + k.two_near / oz;
    d[0] = k.cw * (dx / dz - ox / oz);
    d[1] = k.ch * (dy / dz - oy / oz);
    d[2] = k.neg_two_near / oz;
}

// one row of the packed ray batch (nerf/train_utils.py:143-168)
NH_DEVICE void nh_write_ray_row(float* r, const float* o, const float* d, float near, float far, const float* vsrc) {
    r[0] = o[0];
    r[1] = o[1];
    r[2] = o[2];
    r[3] = d[0];
    r[4] = d[1];
    r[5] = d[2];
    r[6] = near;
    r[7] = far;
    if (vsrc) {
        float x = vsrc[0], y = vsrc[1], z = vsrc[2];
        float nrm = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));  // torch's CPU norm(p=2) is this fma chain (bit-exact)
        r[8] = x / nrm;
        r[9] = y / nrm;
        r[10] = z / nrm;
    }
}

