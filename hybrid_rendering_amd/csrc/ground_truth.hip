// GroundTruthPathTracer on MI355X — HIP replacement for src/ground_truth_path_tracer.cpp (render :44-111) and
// shaders/ground_truth/ground_truth_path_trace.{rgen:52-112, rchit:114-142, rmiss:24-32}.  SURVEY.md §8f row 3.
// One wave = one 8x8 pixel tile, one lane = one pixel: jittered primary ray (closest hit), direct lighting at the hit
// (punctual light with soft shadows + one cosine-lobe sky sample, each with a visibility ray), running mean into the
// ping-pong RGBA16F images.  indirect_lighting (rchit:67-108) contributes nothing upstream — its recursive traceRayEXT
// is commented out (rchit:95-105) and p_IndirectPayload.L stays 0.  hr_ground_truth_params.trace_indirect = 1 re-enables
// that call (SURVEY §8f row 3's optional extension): up to max_ray_bounces path segments with Russian roulette.
#include "hr_internal.h"
#include "shading.h"

using namespace hr;

struct GTArgs
{
    float        view_inverse[16], proj_inverse[16];
    hr_light     light;
    const Node8* nodes;
    const TriGPU* tris;
    SceneShading sh;
    CubeMap      sky;
    const uint2* prev;
    uint2*       cur;
    uint32_t*    ray_slots;   // rays per 8x8 tile
    int          w, h, y0, y1, tiles_x, tile_y0;
    uint32_t     num_frames;
    float        roughness_multiplier;
    uint32_t     max_ray_bounces;
    int          trace_indirect;
};

#define GT_MAX_DEPTH 32
// GLSL max(x, y) = (x < y) ? y : x: keeps a NaN first argument (the indirect path can produce 0/0), unlike max2
HR_DEV float glsl_max(float x, float y) { return (x < y) ? y : x; }
// brdf.glsl:96-112
HR_DEV f3 sample_specular_ggx_lobe(f3 n, float alpha, float xi_x, float xi_y)
{
    const float phi = 2.0f * HR_M_PI * xi_x;
    const float ct  = hr_sqrt(__fdiv_rn(1.0f - xi_y, 1.0f + (alpha * alpha - 1.0f) * xi_y));
    const float st  = hr_sqrt(1.0f - ct * ct);
    float s, c;
    det_sincos(phi, s, c);
    const f3 t   = mk3(st * c, st * s, ct);
    const f3 ref = fabsf(dot3(n, mk3(0.0f, 1.0f, 0.0f))) > 0.99f ? mk3(0.0f, 0.0f, 1.0f) : mk3(0.0f, 1.0f, 0.0f);
    const f3 x   = normalize3(cross3(ref, n));
    const f3 y   = cross3(n, x);
    return normalize3(mk3((x.x * t.x + y.x * t.y) + n.x * t.z, (x.y * t.x + y.y * t.y) + n.y * t.z, (x.z * t.x + y.z * t.y) + n.z * t.z));
}

// INDIRECT = false is the reference as shipped (one invocation per pixel: no payload chain to keep)
template <bool INDIRECT>
__global__ __launch_bounds__(64) void k_ground_truth(GTArgs a)
{
    __shared__ uint32_t s_stack[HR_STACK_ENTRIES * 64];
    const int lane = threadIdx.x;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x + a.tile_y0;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    uint32_t  rays = 0;
    if (x < a.w && y >= a.y0 && y < a.y1)
    {
        Rng rng = rng_init((uint32_t)x, (uint32_t)y, a.num_frames);
        const float jx = next_float(rng), jy = next_float(rng);
        const float px = ((float)x + 0.5f) + jx, py = ((float)y + 0.5f) + jy;
        const float tcx = __fdiv_rn(px, (float)a.w) * 2.0f - 1.0f, tcy = __fdiv_rn(py, (float)a.h) * 2.0f - 1.0f;
        const f4 origin = mul_m4(a.view_inverse, 0.0f, 0.0f, 0.0f, 1.0f);
        const f4 target = mul_m4(a.proj_inverse, tcx, tcy, 1.0f, 1.0f);
        const f3 tn     = normalize3(mk3(target.x, target.y, target.z));
        const f4 dir4   = mul_m4(a.view_inverse, tn.x, tn.y, tn.z, 0.0f);
        // The payload chain of the recursive shader, unrolled.  Every nested invocation starts with L = 0 and its L is ADDED to
        // its caller's (p_Payload.L += indirect_lighting()), innermost first: adds[k] is what invocation k adds.  With
        // trace_indirect = 0 (the reference as shipped: the recursive traceRayEXT of rchit:95-105 is commented out) the loop
        // runs once.  trace_indirect = 1 re-enables that call (SURVEY 8f row 3's optional extension), rchit:67-108 verbatim.
        f3       o = mk3(origin.x, origin.y, origin.z), d = mk3(dir4.x, dir4.y, dir4.z);
        f3       T = one3();
        float    t_min = 0.001f;
        f3       adds[INDIRECT ? GT_MAX_DEPTH : 1];
        int      depth = 0;
        uint32_t nn = 0, nt = 0;
        for (;; depth++)
        {
            adds[depth] = mk3(0.0f, 0.0f, 0.0f);
            rays++;
            const HitRec h = trace_closest(a.nodes, a.tris, o, d, t_min, 10000.0f, s_stack, lane);
            if (h.prim < 0)
            {
                const f3 env = a.sky.fetch(d);
                adds[depth]  = depth == 0 ? env : mul3(T, env);   // rmiss:26-34
                break;
            }
            SurfaceHit s = surface_at(a.sh, h);
            s.N = normalize3(s.N);   // rchit:131 normalises fetch_normal()'s result once more (observable with normal maps)
            const float roughness = s.roughness * a.roughness_multiplier;
            const f3 Wo = neg3(d);
            const f3 F0 = mix3(mk3(0.04f, 0.04f, 0.04f), s.albedo, s.metallic);
            const f3 c_diffuse = mix3(mul3(s.albedo, sub3(one3(), F0)), mk3(0.0f, 0.0f, 0.0f), s.metallic);
            const float r1x = next_float(rng), r1y = next_float(rng), r2x = next_float(rng), r2y = next_float(rng);
            f3       Lo = mk3(0.0f, 0.0f, 0.0f);
            const f3 ray_origin = add3(s.P, scale3(s.N, 0.1f));
            {
                f3    Wi;
                float t_max, attenuation;
                fetch_light_shadow(a.light, s.P, s.N, r1x, r1y, Wi, t_max, attenuation);
                const f3 Li = scale3(mk3(a.light.data2[0], a.light.data2[1], a.light.data2[2]), a.light.data0[3]);
                const f3 Wh = normalize3(add3(Wo, Wi));
                if (attenuation > 0.0f)
                {
                    rays++;
                    attenuation = attenuation * (trace_any<false>(a.nodes, a.tris, ray_origin, Wi, 0.01f, t_max, s_stack, lane, nn, nt) ? 0.0f : 1.0f);
                }
                const f3 brdf = evaluate_uber_brdf(c_diffuse, roughness, s.N, F0, Wo, Wh, Wi);
                Lo = add3(Lo, mul3(scale3(mul3(T, brdf), attenuation), Li));
            }
            {
                const f3 Wi = sample_cosine_lobe_n(s.N, r2x, r2y);
                f3       Li = a.sky.fetch(Wi);
                const f3 Wh = normalize3(add3(Wo, Wi));
                rays++;
                Li = scale3(Li, trace_any<false>(a.nodes, a.tris, ray_origin, Wi, 0.01f, 10000.0f, s_stack, lane, nn, nt) ? 0.0f : 1.0f);
                const f3 brdf = evaluate_uber_brdf(c_diffuse, roughness, s.N, F0, Wo, Wh, Wi);
                Lo = add3(Lo, mul3(mul3(T, brdf), Li));
            }
            adds[depth] = Lo;
            if (!INDIRECT || !((uint32_t)(depth + 1) < a.max_ray_bounces) || depth + 1 >= GT_MAX_DEPTH - 1) break;
            // indirect_lighting (rchit:67-108); sample_uber_brdf takes the RNG BY VALUE (brdf.glsl:146): the lobe sample re-uses
            // the numbers the Russian roulette and the next bounce draw
            Rng copy = rng;
            const float rvx = next_float(copy), rvy = next_float(copy), rvz = next_float(copy);
            const float alpha = roughness * roughness;
            f3 Wi, Wh;
            if (rvx < 0.5f)
            {
                Wh = sample_specular_ggx_lobe(s.N, alpha, rvy, rvz);
                const f3 I = neg3(Wo);
                Wi = roughness < 0.05f ? sub3(I, scale3(s.N, 2.0f * dot3(s.N, I))) : sub3(I, scale3(Wh, 2.0f * dot3(Wh, I)));
            }
            else
            {
                Wi = sample_cosine_lobe_n(s.N, rvy, rvz);
                Wh = normalize3(add3(Wo, Wi));
            }
            const float NdotL = glsl_max(dot3(s.N, Wi), 0.0f), NdotH = glsl_max(dot3(s.N, Wh), 0.0f), VdotH = glsl_max(dot3(Wi, Wh), 0.0f);
            const float pd  = __fdiv_rn(NdotL, HR_M_PI);
            const float ps  = __fdiv_rn(D_ggx(NdotH, alpha) * NdotH, glsl_max(HR_EPSILON, 4.0f * VdotH));
            const float pdf = mix1(pd, ps, 0.5f);
            const f3    brdf = evaluate_uber_brdf(c_diffuse, roughness, s.N, F0, Wo, Wh, Wi);
            const float cos_theta = clamp1(dot3(s.N, Wi), 0.0f, 1.0f);
            const f3    tb = mul3(T, scale3(brdf, cos_theta));
            f3 Tn = mk3(__fdiv_rn(tb.x, pdf), __fdiv_rn(tb.y, pdf), __fdiv_rn(tb.z, pdf));
            const float probability = glsl_max(Tn.x, glsl_max(Tn.y, Tn.z));
            if (next_float(rng) > probability) break;
            Tn = scale3(Tn, __fdiv_rn(1.0f, probability));
            T = Tn; o = s.P; d = Wi; t_min = 0.0001f;
        }
        f3 L = adds[depth];
        for (int k = depth - 1; k >= 0; k--) L = add3(adds[k], L);
        const f3 clamped = mk3(L.x < 1.0f ? L.x : 1.0f, L.y < 1.0f ? L.y : 1.0f, L.z < 1.0f ? L.z : 1.0f); // RADIANCE_CLAMP_COLOR
        f3 out = clamped;
        const size_t i = (size_t)y * a.w + x;
        if (a.num_frames != 0u)
        {
            const uint2 q  = a.prev[i];
            const f3    pc = mk3(h2f_lo(q.x), h2f_hi(q.x), h2f_lo(q.y));
            const float n  = (float)a.num_frames;
            out = mk3(pc.x + __fdiv_rn(clamped.x - pc.x, n), pc.y + __fdiv_rn(clamped.y - pc.y, n), pc.z + __fdiv_rn(clamped.z - pc.z, n));
        }
        a.cur[i] = make_uint2(pack_h2(out.x, out.y), pack_h2(out.z, 1.0f));
    }
    for (int o = 32; o > 0; o >>= 1) rays += __shfl_down(rays, o);
    if (lane == 0) a.ray_slots[blockIdx.x] = rays;
}

struct hr_ground_truth
{
    hr_ctx*  ctx = nullptr;
    int      w = 0, h = 0, y0 = 0, y1 = 0, tiles_x = 0;
    DevBuf   image[2], ray_slots;
    uint32_t frame_idx = 0;
    bool     ping_pong = false;
    hipStream_t last_stream = nullptr;
    StageProfiler prof;
};

extern "C" {

void hr_ground_truth_default_params(hr_ground_truth_params* p)
{
    p->max_ray_bounces = 2;         // ground_truth_path_tracer.h:30
    p->roughness_multiplier = 1.0f; // CommonResources::roughness_multiplier
    p->trace_indirect = 0;          // the reference ships the recursive trace commented out (rchit:95-105)
}

hr_status hr_ground_truth_create(hr_ctx* ctx, int32_t width, int32_t height, const hr_band* band, hr_ground_truth** out)
{
    HR_CHECK_ARG(ctx && out && width > 0 && height > 0);
    HR_HIP(hipSetDevice(ctx->device));
    hr_ground_truth* p = new hr_ground_truth();
    p->ctx = ctx; p->w = width; p->h = height; p->y0 = 0; p->y1 = height;
    if (band && band->band_y1 > band->band_y0)
    {
        // pixels are independent: a band needs no halo and no exchange
        p->y0 = band->band_y0; p->y1 = band->band_y1 > height ? height : band->band_y1;
        if ((p->y0 & 7) || p->y0 < 0 || p->y0 >= p->y1) { set_last_error("band_y0 must be a multiple of 8 inside the image"); delete p; return HR_ERR_INVALID_ARG; }
    }
    p->tiles_x = cdiv(width, 8);
    hr_status s;
    for (int i = 0; i < 2; i++)
    {
        if ((s = p->image[i].alloc((size_t)width * height * 8)) != HR_OK) { delete p; return s; }
        HR_HIP(hipMemset(p->image[i].p, 0, p->image[i].bytes));
    }
    if ((s = p->ray_slots.alloc((size_t)p->tiles_x * cdiv(height, 8) * 4)) != HR_OK) { delete p; return s; }
    HR_HIP(hipMemset(p->ray_slots.p, 0, p->ray_slots.bytes));
    *out = p;
    return HR_OK;
}

hr_status hr_ground_truth_destroy(hr_ground_truth* p)
{
    if (!p) return HR_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    delete p;
    return HR_OK;
}

hr_status hr_ground_truth_restart_accumulation(hr_ground_truth* p) { HR_CHECK_ARG(p); p->frame_idx = 0; return HR_OK; }
hr_status hr_ground_truth_set_profiling(hr_ground_truth* p, int32_t e) { HR_CHECK_ARG(p); p->prof.enabled = e != 0; return HR_OK; }
hr_status hr_ground_truth_get_stage_times(hr_ground_truth* p, hr_stage_times* out) { HR_CHECK_ARG(p && out); p->prof.collect(out); return HR_OK; }

hr_status hr_ground_truth_render(hr_ground_truth* p, const hr_scene* scene, const hr_ubo* ubo, const hr_environment* env, const hr_ground_truth_params* prm, void* stream_)
{
    HR_CHECK_ARG(p && scene && ubo && env && prm && env->sky && env->sky_size > 0);
    HR_HIP(hipSetDevice(p->ctx->device));
    hipStream_t st = (hipStream_t)stream_;
    p->last_stream = st;
    p->prof.begin_frame();
    if (p->frame_idx == 0) p->ping_pong = false; // ground_truth_path_tracer.cpp:50-51
    const int read_idx = p->ping_pong ? 1 : 0, write_idx = p->ping_pong ? 0 : 1;
    GTArgs a;
    for (int i = 0; i < 16; i++) { a.view_inverse[i] = ubo->view_inverse[i]; a.proj_inverse[i] = ubo->proj_inverse[i]; }
    a.light = ubo->light;
    a.nodes = (const Node8*)scene->nodes.p; a.tris = (const TriGPU*)scene->tris.p;
    scene_shading_from(scene, a.sh);
    a.sky = CubeMap { (const uint2*)env->sky, env->sky_size };
    a.prev = (const uint2*)p->image[read_idx].p; a.cur = (uint2*)p->image[write_idx].p;
    a.ray_slots = (uint32_t*)p->ray_slots.p;
    a.w = p->w; a.h = p->h; a.y0 = p->y0; a.y1 = p->y1; a.tiles_x = p->tiles_x; a.tile_y0 = p->y0 / 8;
    a.num_frames = p->frame_idx++;
    a.roughness_multiplier = prm->roughness_multiplier;
    a.max_ray_bounces = (uint32_t)prm->max_ray_bounces; a.trace_indirect = prm->trace_indirect ? 1 : 0;
    const int tiles_y = cdiv(p->y1, 8) - a.tile_y0;
    const uint64_t px = (uint64_t)p->w * (p->y1 - p->y0);
    int ev = p->prof.begin("path_trace", st, px * 16);
    if (a.trace_indirect) hipLaunchKernelGGL(k_ground_truth<true>, dim3(p->tiles_x * tiles_y), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(k_ground_truth<false>, dim3(p->tiles_x * tiles_y), dim3(64), 0, st, a);
    p->prof.end(ev, st);
    HR_HIP(hipGetLastError());
    p->ping_pong = !p->ping_pong;
    return HR_OK;
}

// GroundTruthPathTracer::output_ds: the image written by the last render()
hr_status hr_ground_truth_output(hr_ground_truth* p, hr_image_view* v)
{
    HR_CHECK_ARG(p && v);
    const int last_write = p->ping_pong ? 1 : 0; // render() flipped ping_pong after writing !ping_pong
    v->data = p->image[last_write].p; v->width = p->w; v->height = p->h; v->row_pitch_bytes = p->w * 8; v->format = HR_FORMAT_RGBA16F;
    return HR_OK;
}

hr_status hr_ground_truth_ray_count(hr_ground_truth* p, uint64_t* rays)
{
    HR_CHECK_ARG(p && rays);
    HR_HIP(hipStreamSynchronize(p->last_stream));
    std::vector<uint32_t> slots(p->ray_slots.bytes / 4);
    HR_HIP(hipMemcpy(slots.data(), p->ray_slots.p, slots.size() * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    const int t0 = (p->y0 / 8) * p->tiles_x, t1 = cdiv(p->y1, 8) * p->tiles_x;
    for (int i = 0; i < t1 - t0; i++) total += slots[i];
    *rays = total;
    return HR_OK;
}

} // extern "C"
