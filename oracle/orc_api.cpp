// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_api.h).
#include "orc_api.h"
#include "orc_bvh.h"
#include "orc_common.h"

using namespace orc;

extern "C" {

void* orc_scene_create(const float* verts, int n_tris, const float* normals, const uint32_t* tri_material, const uint32_t* tri_mesh_id, const float* materials, int n_materials)
{
    Scene* s = new Scene();
    s->build(verts, n_tris);
    if (normals) s->tri_normals.assign(normals, normals + (size_t)n_tris * 9);
    if (tri_material) s->tri_material.assign(tri_material, tri_material + n_tris);
    if (tri_mesh_id) s->tri_mesh_id.assign(tri_mesh_id, tri_mesh_id + n_tris);
    if (materials) s->materials.assign(materials, materials + (size_t)n_materials * 8);
    return s;
}
// textured materials: uvs [n][3][2], tangents [n][3][3] (nullable), mat_tex [m][6], textures: n_tex RGBA8 images
void orc_scene_set_textures(void* scene, const float* uvs, const float* tangents, const int32_t* mat_tex, int n_materials, int n_tex,
                            const uint8_t* const* tex_rgba, const int32_t* tex_w, const int32_t* tex_h)
{
    Scene*       s = (Scene*)scene;
    const size_t n = s->tris.size();
    if (uvs) s->tri_uvs.assign(uvs, uvs + n * 6);
    if (tangents) s->tri_tangents.assign(tangents, tangents + n * 9);
    s->mat_tex.assign(mat_tex, mat_tex + (size_t)n_materials * 6);
    s->textures.resize(n_tex);
    for (int i = 0; i < n_tex; i++)
    {
        s->textures[i].w = tex_w[i]; s->textures[i].h = tex_h[i];
        s->textures[i].rgba.assign(tex_rgba[i], tex_rgba[i] + (size_t)tex_w[i] * tex_h[i] * 4);
    }
}
void orc_instances_flatten(int n_instances, const float* matrices, const uint32_t* first_tri, const uint32_t* mesh_tri_base, const uint32_t* n_tris,
                           const float* mesh_positions, const float* mesh_normals, float* out_positions, float* out_normals)
{
    for (int i = 0; i < n_instances; i++)
    {
        const float* m = matrices + (size_t)i * 16;
        for (uint32_t t = 0; t < n_tris[i]; t++)
            for (int v = 0; v < 3; v++)
            {
                const float* p = mesh_positions + ((size_t)mesh_tri_base[i] + t) * 9 + v * 3;
                float*       o = out_positions + ((size_t)first_tri[i] + t) * 9 + v * 3;
                for (int r = 0; r < 3; r++) o[r] = ((m[r] * p[0] + m[4 + r] * p[1]) + m[8 + r] * p[2]) + m[12 + r] * 1.0f;
                if (mesh_normals && out_normals)
                {
                    const float* n = mesh_normals + ((size_t)mesh_tri_base[i] + t) * 9 + v * 3;
                    float*       q = out_normals + ((size_t)first_tri[i] + t) * 9 + v * 3;
                    for (int r = 0; r < 3; r++) q[r] = (m[r] * n[0] + m[4 + r] * n[1]) + m[8 + r] * n[2];
                }
            }
    }
}
void orc_scene_set_instances(void* scene, int n_instances, const float* matrices, const uint32_t* first_tri, const uint32_t* mesh_tri_base, const uint32_t* mesh_id,
                             const uint32_t* n_tris, int n_mesh_tris, const float* mesh_positions, const float* mesh_normals, const uint32_t* mesh_material,
                             const float* mesh_uvs, const float* mesh_tangents)
{
    Scene* s = (Scene*)scene;
    s->instances.resize((size_t)n_instances);
    s->tri_instance.assign(s->tris.size(), 0u);
    for (int i = 0; i < n_instances; i++)
    {
        Scene::Instance& r = s->instances[(size_t)i];
        for (int k = 0; k < 16; k++) r.m[k] = matrices[(size_t)i * 16 + k];
        r.first_tri = first_tri[i]; r.mesh_tri_base = mesh_tri_base[i]; r.mesh_id = mesh_id[i]; r.n_tris = n_tris[i];
        for (uint32_t t = 0; t < r.n_tris; t++) s->tri_instance[(size_t)r.first_tri + t] = (uint32_t)i;
    }
    const size_t n = (size_t)n_mesh_tris;
    s->mesh_positions.assign(mesh_positions, mesh_positions + n * 9);
    if (mesh_normals) s->mesh_normals.assign(mesh_normals, mesh_normals + n * 9);
    if (mesh_material) s->mesh_material.assign(mesh_material, mesh_material + n);
    if (mesh_uvs) s->mesh_uvs.assign(mesh_uvs, mesh_uvs + n * 6);
    if (mesh_tangents) s->mesh_tangents.assign(mesh_tangents, mesh_tangents + n * 9);
}
void orc_scene_destroy(void* scene) { delete (Scene*)scene; }
int  orc_scene_num_nodes(const void* scene) { return (int)((const Scene*)scene)->nodes.size(); }

// single-ray entry points: the ray-query / traceRay mock of oracle/refshim calls these through a function pointer
int orc_any_hit_one(void* scene, const float* o, const float* d, float t_min, float t_max)
{
    return ((const Scene*)scene)->any_hit(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), t_min, t_max) ? 1 : 0;
}
int orc_closest_hit_one(void* scene, const float* o, const float* d, float t_min, float t_max, float* tuv)
{
    Hit h = ((const Scene*)scene)->closest_hit(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), t_min, t_max);
    tuv[0] = h.t; tuv[1] = h.u; tuv[2] = h.v;
    return h.prim;
}

void orc_any_hit_batch(const void* scene, int n, const float* rays, uint8_t* out, int brute_force, uint64_t* stats)
{
    const Scene& s = *(const Scene*)scene;
    if (stats)
    {
        traversal_stats() = TraversalStats();
        // serial, instrumented
        for (int i = 0; i < n; i++)
        {
            const float* r = rays + (size_t)i * 8;
            out[i] = s.any_hit(v3(r[0], r[1], r[2]), v3(r[4], r[5], r[6]), r[7], r[3]) ? 1 : 0;
        }
        stats[0] = traversal_stats().nodes;
        stats[1] = traversal_stats().tris;
        return;
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++)
    {
        const float* r = rays + (size_t)i * 8;
        vec3 o = v3(r[0], r[1], r[2]), d = v3(r[4], r[5], r[6]);
        bool h = brute_force ? s.any_hit_brute(o, d, r[7], r[3]) : s.any_hit(o, d, r[7], r[3]);
        out[i] = h ? 1 : 0;
    }
}

void orc_closest_hit_batch(const void* scene, int n, const float* rays, float* out_tuv, int32_t* out_prim, int brute_force)
{
    const Scene& s = *(const Scene*)scene;
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++)
    {
        const float* r = rays + (size_t)i * 8;
        vec3 o = v3(r[0], r[1], r[2]), d = v3(r[4], r[5], r[6]);
        Hit  h = brute_force ? s.closest_hit_brute(o, d, r[7], r[3]) : s.closest_hit(o, d, r[7], r[3]);
        out_tuv[(size_t)i * 3 + 0] = h.t;
        out_tuv[(size_t)i * 3 + 1] = h.u;
        out_tuv[(size_t)i * 3 + 2] = h.v;
        out_prim[i]                = h.prim;
    }
}

// ---------------------------------------------------------------------------------------
// G-buffer synthesis by primary-ray casting.  Output conventions follow g_buffer.frag:86-112:
//   GB1 RGBA8  = albedo.rgb, metallic
//   GB2 RGBA16F = octahedral normal.xy, motion (prev_uv - cur_uv)          (:47-67,103)
//   GB3 RGBA16F = roughness, curvature, mesh id, linear z (= clip z)         (:106-111)
//   depth F32  = clip.z / clip.w, sky = 1.0;  GB3 sky = (0,0,0,-1)           (g_buffer.cpp:88,96)
// Test tooling, not a restatement of a reference shader (the reference rasterises).
static inline vec3 tri_normal_at(const Scene& s, int prim, float b0, float b1, float b2)
{
    if (!s.tri_normals.empty())
    {
        const float* n = &s.tri_normals[(size_t)prim * 9];
        return v3(n[0] * b0 + n[3] * b1 + n[6] * b2, n[1] * b0 + n[4] * b1 + n[7] * b2, n[2] * b0 + n[5] * b1 + n[8] * b2);
    }
    const Tri& t = s.tris[prim];
    return normalize(cross(t.v1 - t.v0, t.v2 - t.v0));
}

static inline bool plane_bary(const Tri& t, vec3 o, vec3 d, float* b0, float* b1, float* b2)
{
    vec3  e1 = t.v1 - t.v0, e2 = t.v2 - t.v0;
    vec3  n  = cross(e1, e2);
    float dn = dot(n, d);
    if (dn == 0.0f) return false;
    float tt = dot(n, t.v0 - o) / dn;
    vec3  p  = (o + d * tt) - t.v0;
    float d11 = dot(e1, e1), d12 = dot(e1, e2), d22 = dot(e2, e2), p1 = dot(p, e1), p2 = dot(p, e2);
    float den = d11 * d22 - d12 * d12;
    if (den == 0.0f) return false;
    *b1 = (d22 * p1 - d12 * p2) / den;
    *b2 = (d11 * p2 - d12 * p1) / den;
    *b0 = 1.0f - *b1 - *b2;
    return true;
}

void orc_gbuffer_raycast(const void* scene_, const void* ubo_, int w, int h, uint8_t* gb1, uint16_t* gb2, uint16_t* gb3, float* depth)
{
    const Scene& s   = *(const Scene*)scene_;
    const UBO&   ubo = *(const UBO*)ubo_;
    const vec3   cam = v3(ubo.cam_pos[0], ubo.cam_pos[1], ubo.cam_pos[2]);
    auto pixel_dir = [&](float px, float py) {
        float u = px / (float)w, v = py / (float)h;
        vec3  far_p = world_position_from_depth(u, v, 1.0f, ubo.view_proj_inverse);
        return normalize(far_p - cam);
    };
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            size_t   i  = (size_t)y * w + x;
            uint8_t* o1 = gb1 + i * 4;
            uint16_t *o2 = gb2 + i * 4, *o3 = gb3 + i * 4;
            vec3 d  = pixel_dir((float)x + 0.5f, (float)y + 0.5f);
            Hit  hit = s.closest_hit(cam, d, 0.0f, 1.0e30f);
            if (hit.prim < 0)
            {
                o1[0] = o1[1] = o1[2] = o1[3] = 0;
                o2[0] = o2[1] = o2[2] = o2[3] = 0;
                o3[0] = o3[1] = o3[2] = 0;
                o3[3] = f32_to_f16(-1.0f);
                depth[i] = 1.0f;
                continue;
            }
            vec3  P    = cam + d * hit.t;
            vec4  clip = mul(ubo.view_proj, vec4 { P.x, P.y, P.z, 1.0f });
            vec4  pclip = mul(ubo.prev_view_proj, vec4 { P.x, P.y, P.z, 1.0f });
            float b0 = 1.0f - hit.u - hit.v;
            vec3  nI = tri_normal_at(s, hit.prim, b0, hit.u, hit.v); // un-normalised interpolated normal
            vec3  n  = normalize(nI);
            if (dot(n, d) > 0.0f) n = -n; // face the viewer (two-sided)
            // curvature = sqrt(max(|dFdx N|^2, |dFdy N|^2)) with attribute extrapolation over the triangle plane
            float curvature = 0.0f;
            {
                float c0, c1, c2;
                vec3  dxv = v3(0, 0, 0), dyv = v3(0, 0, 0);
                if (plane_bary(s.tris[hit.prim], cam, pixel_dir((float)x + 1.5f, (float)y + 0.5f), &c0, &c1, &c2)) dxv = tri_normal_at(s, hit.prim, c0, c1, c2) - nI;
                if (plane_bary(s.tris[hit.prim], cam, pixel_dir((float)x + 0.5f, (float)y + 1.5f), &c0, &c1, &c2)) dyv = tri_normal_at(s, hit.prim, c0, c1, c2) - nI;
                if (s.tri_normals.empty()) { dxv = v3(0, 0, 0); dyv = v3(0, 0, 0); }
                curvature = std::sqrt(fmax2(dot(dxv, dxv), dot(dyv, dyv)));
            }
            vec2  oct = direction_to_octohedral(n);
            float cx = clip.x / clip.w * 0.5f + 0.5f, cy = clip.y / clip.w * 0.5f + 0.5f;
            float px = pclip.x / pclip.w * 0.5f + 0.5f, py = pclip.y / pclip.w * 0.5f + 0.5f;
            uint32_t mat = s.tri_material.empty() ? 0u : s.tri_material[hit.prim];
            float albedo[3] = { 0.8f, 0.8f, 0.8f }, metallic = 0.0f, roughness = 0.5f;
            if (!s.materials.empty())
            {
                const float* m = &s.materials[(size_t)mat * 8];
                albedo[0] = m[0]; albedo[1] = m[1]; albedo[2] = m[2]; metallic = m[3]; roughness = m[4];
            }
            for (int c = 0; c < 3; c++) o1[c] = (uint8_t)(clampf(albedo[c], 0.0f, 1.0f) * 255.0f + 0.5f);
            o1[3] = (uint8_t)(clampf(metallic, 0.0f, 1.0f) * 255.0f + 0.5f);
            o2[0] = f32_to_f16(oct.x); o2[1] = f32_to_f16(oct.y);
            o2[2] = f32_to_f16(px - cx); o2[3] = f32_to_f16(py - cy);
            float mesh_id = s.tri_mesh_id.empty() ? 0.0f : (float)s.tri_mesh_id[hit.prim];
            o3[0] = f32_to_f16(fmax2(roughness, 0.1f)); o3[1] = f32_to_f16(curvature);
            o3[2] = f32_to_f16(mesh_id); o3[3] = f32_to_f16(clip.z);
            float dd = clip.z / clip.w;
            depth[i] = dd >= 1.0f ? 0.99999994f : dd;
        }
}

// ---- scalar helpers ---------------------------------------------------------------------
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float    orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
void     orc_sincos(float x, float* s, float* c) { det_sincos(x, s, c); }
float    orc_exp(float x) { return det_exp(x); }
float    orc_log(float x) { return det_log(x); }
float    orc_pow(float x, float y) { return det_pow_auto(x, y); }
void     orc_oct_decode(float ex, float ey, float* out3)
{
    vec3 v = octohedral_to_direction(ex, ey);
    out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
void orc_oct_encode(const float* n3, float* out2)
{
    vec2 v = direction_to_octohedral(v3(n3[0], n3[1], n3[2]));
    out2[0] = v.x; out2[1] = v.y;
}
float orc_sample_blue_noise(int x, int y, int sample_index, int dim, const uint8_t* sobol, const uint8_t* scrambling_ranking)
{
    BlueNoise bn { sobol, scrambling_ranking };
    return sample_blue_noise(x, y, sample_index, dim, bn);
}
void orc_world_position_from_depth(float u, float v, float depth, const float* view_proj_inverse, float* out3)
{
    mat4 m;
    for (int i = 0; i < 16; i++) m.m[i] = view_proj_inverse[i];
    vec3 p = world_position_from_depth(u, v, depth, m);
    out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}
}
