// hrt_pack.h — host-side repacking of the flat scene's triangle arrays into the
// 16-byte aligned records the kernels fetch with dwordx4 loads (DScene in
// hrt_device.h).  Used by hrt_scene_create (hrt_hip.hip).
#pragma once
#include <vector>

#include "../../include/hrt.h"
#include "hrt_glm.h"

namespace hrt {

//   pos : 3 x float4 per triangle  v0.xyz_ v1.xyz_ v2.xyz_
//   attr: 4 x float4 per triangle  n0.xyz uv0.x | n1.xyz uv0.y | n2.xyz uv1.x | uv1.y uv2.x uv2.y _
//   box : 2 x float4 per triangle  accept-box min.xyz + tri_ref_order code (raw bits) | max.xyz _
inline void pack_triangles(const hrt_flat_scene* f, std::vector<float>& pos, std::vector<float>& attr, std::vector<float>& box) {
    const size_t nt = (size_t)f->n_tris;
    pos.assign(nt * 12, 0.0f); attr.assign(nt * 16, 0.0f); box.assign(nt * 8, 0.0f);
    for (size_t i = 0; i < nt; ++i) {
        const float* p = f->tri_pos + 9 * i; const float* n = f->tri_nrm + 9 * i; const float* uv = f->tri_uv + 6 * i;
        for (int k = 0; k < 3; ++k) { pos[12 * i + 4 * k] = p[3 * k]; pos[12 * i + 4 * k + 1] = p[3 * k + 1]; pos[12 * i + 4 * k + 2] = p[3 * k + 2]; }
        float* a = &attr[16 * i];
        a[0] = n[0]; a[1] = n[1]; a[2] = n[2]; a[3] = uv[0];
        a[4] = n[3]; a[5] = n[4]; a[6] = n[5]; a[7] = uv[1];
        a[8] = n[6]; a[9] = n[7]; a[10] = n[8]; a[11] = uv[2];
        a[12] = uv[3]; a[13] = uv[4]; a[14] = uv[5];
        float* b = &box[8 * i];
        if (f->tri_box) {
            const float* s = f->tri_box + 6 * i;
            b[0] = s[0]; b[1] = s[1]; b[2] = s[2]; b[4] = s[3]; b[5] = s[4]; b[6] = s[5];
        } else {  // triangle.cpp:133-151
            for (int c = 0; c < 3; ++c) {
                float mn = gmin(gmin(p[c], p[3 + c]), p[6 + c]);
                float mx = gmax(gmax(p[c], p[3 + c]), p[6 + c]);
                b[c] = mn - 0.0001f; b[4 + c] = mx + 0.0001f;
            }
        }
        const uint32_t ord = f->tri_ref_order ? f->tri_ref_order[i] : ((uint32_t)i << 1);
        b[3] = u2f(ord);
    }
}

}  // namespace hrt
