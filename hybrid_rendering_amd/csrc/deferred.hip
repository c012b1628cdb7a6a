// DeferredShading composite on MI355X — HIP replacement for the shading part of src/deferred_shading.cpp
// (render_shading :690-760) and shaders/deferred.frag:177-205 (+ evaluate_sh9_irradiance :115-141,
// indirect_lighting :151-173, lighting.glsl direct_lighting without RAY_TRACING).  SURVEY.md §8f row 1.
#include "hr_internal.h"
#include "shading.h"

using namespace hr;

struct DeferredArgs
{
    float           vpi[16];
    float           cam[3];
    hr_light        light;
    const uint32_t* gb1;
    const uint2*    gb2;
    const uint2*    gb3;
    const float*    depth;
    const uint16_t* shadow; int shadow_ch;
    const uint16_t* ao;     int ao_ch;
    const uint2*    refl;
    const uint2*    gi;
    const uint2*    prefiltered; int pre_size, pre_levels;
    const uint32_t* lut;    int lut_size;
    float           sh9[9][4];
    uint2*          out;
    int             w, h, flags;
    // render_skybox (deferred_shading.cpp:734-789, skybox.vert/.frag): sky texels (depth == 1) take the cubemap colour
    const uint2*    sky; int sky_size;      // null: not drawn
    float           view_inverse[16], proj_inverse[16];
};

__global__ __launch_bounds__(256) void k_deferred(DeferredArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.w || y >= a.h) return;
    const size_t i = (size_t)y * a.w + x;
    const float tu = __fdiv_rn((float)x + 0.5f, (float)a.w), tv = __fdiv_rn((float)y + 0.5f, (float)a.h);
    const uint32_t g1 = a.gb1[i];
    const f3    albedo   = mk3(__fdiv_rn((float)(g1 & 0xffu), 255.0f), __fdiv_rn((float)((g1 >> 8) & 0xffu), 255.0f), __fdiv_rn((float)((g1 >> 16) & 0xffu), 255.0f));
    const float metallic = __fdiv_rn((float)(g1 >> 24), 255.0f);
    const uint2 g2 = a.gb2[i], g3 = a.gb3[i];
    const float roughness = h2f_lo(g3.x);
    const f3    P = world_pos_from_depth(tu, tv, a.depth[i], a.vpi);
    const float visibility = (a.flags & 1) ? h2f(a.shadow[i * a.shadow_ch]) : 1.0f;
    const float aov        = (a.flags & 2) ? h2f(a.ao[i * a.ao_ch]) : 1.0f;
    const f3    N  = oct_decode(h2f_lo(g2.x), h2f_hi(g2.x));
    const f3    Wo = normalize3(sub3(mk3(a.cam[0], a.cam[1], a.cam[2]), P));
    const f3    F0 = mix3(mk3(0.04f, 0.04f, 0.04f), albedo, metallic);
    const f3    c_diffuse = mix3(mul3(albedo, sub3(one3(), F0)), mk3(0.0f, 0.0f, 0.0f), metallic);
    f3 Lo = mk3(0.0f, 0.0f, 0.0f);
    {
        f3    Li, Wi, Wh;
        float t_max, attenuation;
        fetch_light_hard(a.light, Wo, P, N, Li, Wi, Wh, t_max, attenuation);
        const f3 brdf = evaluate_uber_brdf(c_diffuse, roughness, N, F0, Wo, Wh, Wi);
        Lo = add3(Lo, scale3(mul3(scale3(mul3(one3(), brdf), attenuation), Li), visibility));
    }
    {
        const f3    I   = neg3(Wo);
        const f3    R   = sub3(I, scale3(N, 2.0f * dot3(N, I)));
        const float ndv = max2(dot3(N, Wo), 0.0f);
        const f3    F   = fresnel_schlick_roughness(ndv, F0, roughness);
        const f3    kD  = scale3(sub3(one3(), F), 1.0f - metallic);
        f3 irradiance;
        if (a.flags & 8) { const uint2 q = a.gi[i]; irradiance = mk3(h2f_lo(q.x), h2f_hi(q.x), h2f_lo(q.y)); }
        else
        {
            const float Pi = 3.141592654f, A0 = Pi, A1 = __fdiv_rn(2.0f * Pi, 3.0f), A2 = Pi * 0.25f;
            float c[9];
            c[0] = 0.282095f;
            c[1] = -0.488603f * N.y;
            c[2] = 0.488603f * N.z;
            c[3] = -0.488603f * N.x;
            c[4] = 1.092548f * N.x * N.y;
            c[5] = -1.092548f * N.y * N.z;
            c[6] = 0.315392f * (3.0f * N.z * N.z - 1.0f);
            c[7] = -1.092548f * N.x * N.z;
            c[8] = 0.546274f * (N.x * N.x - N.y * N.y);
            c[0] *= A0; c[1] *= A1; c[2] *= A1; c[3] *= A1; c[4] *= A2; c[5] *= A2; c[6] *= A2; c[7] *= A2; c[8] *= A2;
            f3 col = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < 9; k++) col = add3(col, scale3(mk3(a.sh9[k][0], a.sh9[k][1], a.sh9[k][2]), c[k]));
            col = mk3(max2(0.0f, col.x), max2(0.0f, col.y), max2(0.0f, col.z));
            irradiance = div3s(col, Pi);
        }
        const f3 diffuse = mul3(irradiance, c_diffuse);
        f3 pre;
        if (a.flags & 4) { const uint2 q = a.refl[i]; pre = mk3(h2f_lo(q.x), h2f_hi(q.x), h2f_lo(q.y)); }
        else
        {
            int level = (int)floorf(roughness * 4.0f + 0.5f);
            level     = level < 0 ? 0 : (level > a.pre_levels - 1 ? a.pre_levels - 1 : level);
            size_t off = 0;
            for (int l = 0; l < level; l++) off += (size_t)6 * (a.pre_size >> l) * (a.pre_size >> l);
            CubeMap cm { a.prefiltered + off, a.pre_size >> level };
            pre = cm.fetch(R);
        }
        int ix = (int)floorf(ndv * (float)a.lut_size), iy = (int)floorf(roughness * (float)a.lut_size);
        ix = ix < 0 ? 0 : (ix > a.lut_size - 1 ? a.lut_size - 1 : ix);
        iy = iy < 0 ? 0 : (iy > a.lut_size - 1 ? a.lut_size - 1 : iy);
        const uint32_t q = a.lut[(size_t)iy * a.lut_size + ix];
        const float bx = h2f_lo(q), by = h2f_hi(q);
        const f3 specular = scale3(mul3(pre, add3(scale3(F, bx), mk3(by, by, by))), 2.0f);
        Lo = add3(Lo, scale3(add3(mul3(kD, diffuse), specular), aov));
    }
    if (a.sky && a.depth[i] == 1.0f)
    {
        // The skybox cube is drawn after the shading with depth test LEQUAL at z = w: it covers exactly the texels the
        // G-buffer left at depth 1.  The fragment's interpolated cube position lies on the ray through the pixel centre
        // (the rasteriser's own interpolation is not reproducible), so the lookup direction is pinned to that ray, computed as
        // the reference computes a pixel's ray elsewhere (ground_truth_path_trace.rgen:70-72); NEAREST cube fetch (contract).
        const f4 target = mul_m4(a.proj_inverse, tu * 2.0f - 1.0f, tv * 2.0f - 1.0f, 1.0f, 1.0f);
        const f3 tn     = normalize3(mk3(target.x, target.y, target.z));
        const f4 dir    = mul_m4(a.view_inverse, tn.x, tn.y, tn.z, 0.0f);
        const CubeMap cm { a.sky, a.sky_size };
        const f3 env = cm.fetch(mk3(dir.x, dir.y, dir.z));
        a.out[i] = make_uint2(pack_h2(env.x, env.y), pack_h2(env.z, 1.0f));
        return;
    }
    a.out[i] = make_uint2(pack_h2(Lo.x, Lo.y), pack_h2(Lo.z, 1.0f));
}

struct hr_deferred
{
    hr_ctx* ctx = nullptr;
    int     w = 0, h = 0;
    DevBuf  out;
};

extern "C" {

void hr_deferred_default_params(hr_deferred_params* p)
{
    p->use_ray_traced_shadows = p->use_ray_traced_ao = p->use_ray_traced_reflections = p->use_ddgi = 1;
    for (int k = 0; k < 9; k++)
        for (int c = 0; c < 4; c++) p->irradiance_sh9[k][c] = 0.0f;
    p->draw_skybox = 1;   // DeferredShading::render = render_shading + render_skybox (deferred_shading.cpp:58-70)
}

hr_status hr_deferred_create(hr_ctx* ctx, int32_t width, int32_t height, hr_deferred** out)
{
    HR_CHECK_ARG(ctx && out && width > 0 && height > 0);
    HR_HIP(hipSetDevice(ctx->device));
    hr_deferred* p = new hr_deferred();
    p->ctx = ctx; p->w = width; p->h = height;
    hr_status s = p->out.alloc((size_t)width * height * 8);
    if (s != HR_OK) { delete p; return s; }
    *out = p;
    return HR_OK;
}

hr_status hr_deferred_destroy(hr_deferred* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}

static bool view_ok(const hr_image_view* v, int w, int h) { return v && v->data && v->width == w && v->height == h; }
static int  view_channels(const hr_image_view* v) { return v->format == HR_FORMAT_R16F ? 1 : (v->format == HR_FORMAT_RG16F ? 2 : 4); }

hr_status hr_deferred_render(hr_deferred* p, const hr_frame_inputs* in, const hr_environment* env, const hr_image_view* shadow,
                             const hr_image_view* ao, const hr_image_view* reflections, const hr_image_view* gi,
                             const hr_deferred_params* prm, void* stream)
{
    HR_SCOPED_SAMPLE("Deferred Shading");
    HR_CHECK_ARG(p && in && env && prm);
    const hr_gbuffer_level& g = in->cur_full.gb2 ? in->cur_full : in->cur;
    HR_CHECK_ARG(g.gb1 && g.gb2 && g.gb3 && g.depth && g.width == p->w && g.height == p->h);
    HR_CHECK_ARG(env->brdf_lut && env->brdf_lut_size > 0);
    DeferredArgs a;
    a.flags = (prm->use_ray_traced_shadows ? 1 : 0) | (prm->use_ray_traced_ao ? 2 : 0) | (prm->use_ray_traced_reflections ? 4 : 0) | (prm->use_ddgi ? 8 : 0);
    if (a.flags & 1) HR_CHECK_ARG(view_ok(shadow, p->w, p->h));
    if (a.flags & 2) HR_CHECK_ARG(view_ok(ao, p->w, p->h));
    if (a.flags & 4) HR_CHECK_ARG(view_ok(reflections, p->w, p->h) && reflections->format == HR_FORMAT_RGBA16F);
    else HR_CHECK_ARG(env->prefiltered && env->prefiltered_levels > 0);
    if (a.flags & 8) HR_CHECK_ARG(view_ok(gi, p->w, p->h) && gi->format == HR_FORMAT_RGBA16F);
    for (int i = 0; i < 16; i++) a.vpi[i] = in->ubo.view_proj_inverse[i];
    for (int i = 0; i < 3; i++) a.cam[i] = in->ubo.cam_pos[i];
    a.light = in->ubo.light;
    a.gb1 = (const uint32_t*)g.gb1; a.gb2 = (const uint2*)g.gb2; a.gb3 = (const uint2*)g.gb3; a.depth = g.depth;
    a.shadow = (a.flags & 1) ? (const uint16_t*)shadow->data : nullptr; a.shadow_ch = (a.flags & 1) ? view_channels(shadow) : 1;
    a.ao = (a.flags & 2) ? (const uint16_t*)ao->data : nullptr; a.ao_ch = (a.flags & 2) ? view_channels(ao) : 1;
    a.refl = (a.flags & 4) ? (const uint2*)reflections->data : nullptr;
    a.gi = (a.flags & 8) ? (const uint2*)gi->data : nullptr;
    a.prefiltered = (const uint2*)env->prefiltered; a.pre_size = env->prefiltered_size; a.pre_levels = env->prefiltered_levels;
    a.lut = (const uint32_t*)env->brdf_lut; a.lut_size = env->brdf_lut_size;
    for (int k = 0; k < 9; k++)
        for (int c = 0; c < 4; c++) a.sh9[k][c] = prm->irradiance_sh9[k][c];
    a.out = (uint2*)p->out.p; a.w = p->w; a.h = p->h;
    a.sky = (prm->draw_skybox && env->sky) ? (const uint2*)env->sky : nullptr; a.sky_size = env->sky_size;
    for (int i = 0; i < 16; i++) { a.view_inverse[i] = in->ubo.view_inverse[i]; a.proj_inverse[i] = in->ubo.proj_inverse[i]; }
    hipLaunchKernelGGL(k_deferred, dim3(cdiv(p->w, 32), cdiv(p->h, 8)), dim3(256), 0, (hipStream_t)stream, a);
    HR_HIP(hipGetLastError());
    return HR_OK;
}

hr_status hr_deferred_output(hr_deferred* p, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    v->data = p->out.p; v->width = p->w; v->height = p->h; v->row_pitch_bytes = p->w * 8; v->format = HR_FORMAT_RGBA16F;
    return HR_OK;
}

} // extern "C"
