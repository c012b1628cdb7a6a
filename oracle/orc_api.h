// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points of the CPU oracle (loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product).
// Pinned against the reference's own shaders run on the CPU (oracle/refshim, tests/test_ref_shaders.py); upstream has no
// tests / golden vectors of its own.
#pragma once
#include <cstdint>

extern "C" {

// ---- scene -----------------------------------------------------------------------------
// verts: [n][3][3] world-space positions.  normals (nullable): [n][3][3] vertex normals.
// tri_material / tri_mesh_id (nullable): per triangle.  materials (nullable): [m][8] =
// albedo rgb, metallic, roughness, emissive rgb.
void*    orc_scene_create(const float* verts, int n_tris, const float* normals, const uint32_t* tri_material, const uint32_t* tri_mesh_id, const float* materials, int n_materials);
void     orc_scene_set_textures(void* scene, const float* uvs, const float* tangents, const int32_t* mat_tex, int n_materials, int n_tex,
                                const uint8_t* const* tex_rgba, const int32_t* tex_w, const int32_t* tex_h);
// Instanced scenes (scene_descriptor_set.glsl:30-34, :150-160; main.cpp:74 rebuilds the TLAS every frame).
// orc_instances_flatten: world vertices = model_matrix * vec4(p, 1), rows summed left to right ((m0 x + m1 y) + m2 z) + m3 (transform_vertex :155);
// world vertex normals (G-buffer synthesis only) = mat3(model_matrix) * n, not normalised.  Instance i's triangles follow instance i - 1's.
// matrices [I][16] column-major; first_tri / mesh_tri_base / n_tris [I]; mesh_* concatenated per-mesh arrays; out_* [sum n_tris][3][3].
void     orc_instances_flatten(int n_instances, const float* matrices, const uint32_t* first_tri, const uint32_t* mesh_tri_base, const uint32_t* n_tris,
                               const float* mesh_positions, const float* mesh_normals, float* out_positions, float* out_normals);
// after orc_scene_create over the flattened vertices: the hit shading interpolates the object-space attributes, then applies the matrix
// (interpolated_vertex + transform_vertex).  mesh_normals / mesh_material / mesh_uvs / mesh_tangents nullable.
void     orc_scene_set_instances(void* scene, int n_instances, const float* matrices, const uint32_t* first_tri, const uint32_t* mesh_tri_base, const uint32_t* mesh_id,
                                 const uint32_t* n_tris, int n_mesh_tris, const float* mesh_positions, const float* mesh_normals, const uint32_t* mesh_material,
                                 const float* mesh_uvs, const float* mesh_tangents);
void     orc_scene_destroy(void* scene);
int      orc_scene_num_nodes(const void* scene);
// rays: [n][8] = origin xyz, t_max, dir xyz, t_min.  out: [n] uint8 (1 = occluded).
int      orc_any_hit_one(void* scene, const float* o, const float* d, float t_min, float t_max);
int      orc_closest_hit_one(void* scene, const float* o, const float* d, float t_min, float t_max, float* tuv /*[3]*/);
void     orc_any_hit_batch(const void* scene, int n, const float* rays, uint8_t* out, int brute_force, uint64_t* stats /*[2] nodes,tris; nullable*/);
// out: [n][4] float = t, u, v, prim (as float bits of int32; -1 miss)
void     orc_closest_hit_batch(const void* scene, int n, const float* rays, float* out_tuv, int32_t* out_prim, int brute_force);

// ---- G-buffer synthesis (test tooling; mirrors g_buffer.frag:86-112 output conventions) --
void     orc_gbuffer_raycast(const void* scene, const void* ubo, int w, int h, uint8_t* gb1, uint16_t* gb2, uint16_t* gb3, float* depth);

// ---- shadows ----------------------------------------------------------------------------
void orc_shadows_ray_trace(const void* scene, const void* ubo, int w, int h, const float* depth, const uint16_t* gb2,
                           const uint8_t* sobol, const uint8_t* scrambling_ranking, float bias, uint32_t num_frames,
                           uint32_t* mask, uint64_t* rays_out);
void orc_shadows_gen_rays(const void* ubo, int w, int h, const float* depth, const uint16_t* gb2, const uint8_t* sobol,
                          const uint8_t* scrambling_ranking, float bias, uint32_t num_frames, float* rays);
void orc_shadows_temporal(const void* ubo, int w, int h, const uint32_t* mask, const float* depth, const uint16_t* gb2,
                          const uint16_t* gb3, const float* prev_depth, const uint16_t* prev_gb2, const uint16_t* prev_gb3,
                          const uint16_t* hist_vis_var, const uint16_t* hist_moments, float alpha, float moments_alpha,
                          uint16_t* out_vis_var, uint16_t* out_moments, uint8_t* tile_class);
void orc_shadows_atrous(int w, int h, const uint16_t* in_vis_var, const uint16_t* gb2, const uint16_t* gb3,
                        const uint8_t* tile_class, int radius, int step_size, float phi_visibility, float phi_normal,
                        float sigma_depth, float power, uint16_t* out_vis_var);
void orc_upsample(int W, int H, int w, int h, const uint16_t* gb2_full, const uint16_t* gb3_full, const uint16_t* gb2_mip,
                  const uint16_t* gb3_mip, const uint16_t* in_lowres, int in_channels, int channels, float sky_value, float power,
                  uint16_t* out_full);

// ---- ambient occlusion ---------------------------------------------------------------------
void orc_ao_ray_trace(const void* scene, const void* ubo, int w, int h, const float* depth, const uint16_t* gb2, const uint8_t* sobol,
                      const uint8_t* scrambling_ranking, float bias, float ray_length, uint32_t num_frames, int spp, uint32_t* mask, uint64_t* rays_out);
void orc_ao_temporal(const void* ubo, int w, int h, int spp, const uint32_t* mask, const float* depth, const uint16_t* gb2, const uint16_t* gb3,
                     const float* prev_depth, const uint16_t* prev_gb2, const uint16_t* prev_gb3, const uint16_t* hist_ao,
                     const uint16_t* hist_len, float alpha, uint16_t* out_ao, uint16_t* out_len, uint8_t* tile_class);
void orc_ao_blur(int w, int h, const uint16_t* in_ao, const float* depth, const uint16_t* gb2, const uint8_t* tile_class, const float* zbp,
                 int dir_x, int dir_y, int radius, uint16_t* out_ao);

// ---- DDGI -----------------------------------------------------------------------------------
// ddgi: 88-byte DDGIUniforms (ddgi.cpp:14-32).  orientation: column-major 3x3 probe-ray rotation.
// sky: [6][S][S][4] fp16 cubemap (+X -X +Y -Y +Z -Z).
void orc_ddgi_ray_trace(const void* scene, const void* ubo, const void* ddgi, const float* orientation, uint32_t num_frames,
                        int infinite_bounces, float gi_intensity, const uint16_t* sky, int sky_size, const uint16_t* prev_irradiance,
                        const uint16_t* prev_depth, uint16_t* radiance, uint16_t* direction_distance, uint64_t* rays_out);
void orc_ddgi_probe_update(const void* ddgi, int depth_probe, int first_frame, const uint16_t* radiance, const uint16_t* direction_distance,
                           const uint16_t* prev_atlas, uint16_t* out_atlas);
void orc_ddgi_border_update(const void* ddgi, int depth_probe, uint16_t* atlas);
void orc_ddgi_sample_probe_grid(const void* ubo, const void* ddgi, int w, int h, const float* depth, const uint16_t* gb2, float gi_intensity,
                                const uint16_t* irradiance, const uint16_t* depth_atlas, uint16_t* out);

// ---- reflections -----------------------------------------------------------------------------
struct orc_refl_trace_params
{
    float    bias, trim;
    uint32_t num_frames;
    int      sample_gi, approximate_with_ddgi;
    float    gi_intensity, rough_ddgi_intensity, ibl_indirect_specular_intensity;
};
void orc_reflections_ray_trace(const void* scene, const void* ubo, const void* ddgi, int w, int h, const float* depth, const uint16_t* gb2,
                               const uint16_t* gb3, const uint8_t* sobol, const uint8_t* scrambling_ranking, const orc_refl_trace_params* prm,
                               const uint16_t* sky, int sky_size, const uint16_t* prefiltered, int pre_size, int pre_levels, const uint16_t* lut,
                               int lut_size, const uint16_t* irradiance, const uint16_t* depth_atlas, uint16_t* out, uint64_t* rays_out);
void orc_reflections_temporal(const void* ubo, int w, int h, const uint16_t* input, const float* depth, const uint16_t* gb2, const uint16_t* gb3,
                              const float* prev_depth, const uint16_t* prev_gb2, const uint16_t* prev_gb3, const uint16_t* hist_color,
                              const uint16_t* hist_moments, const float* camera_delta, float alpha, float moments_alpha, int approximate_with_ddgi,
                              uint16_t* out_color, uint16_t* out_moments, uint8_t* tile_class);
void orc_reflections_atrous(int w, int h, const uint16_t* in_color, const float* depth, const uint16_t* gb2, const uint16_t* gb3,
                            const uint8_t* tile_class, int radius, int step_size, float phi_color, float phi_normal, float sigma_depth,
                            int approximate_with_ddgi, uint16_t* out_color);

// ---- deferred composite ---------------------------------------------------
void orc_deferred_shade(const void* ubo, int w, int h, const uint8_t* gb1, const uint16_t* gb2, const uint16_t* gb3, const float* depth,
                        const uint16_t* shadow, int shadow_channels, const uint16_t* ao, int ao_channels, const uint16_t* reflections,
                        const uint16_t* gi, int flags, const float* sh9, const uint16_t* prefiltered, int pre_size, int pre_levels,
                        const uint16_t* lut, int lut_size, uint16_t* out);

// ---- scalar helpers exported for known-answer tests ---------------------------------------
uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);
void     orc_sincos(float x, float* s, float* c);
float    orc_exp(float x);
float    orc_log(float x);
float    orc_pow(float x, float y);
void     orc_oct_decode(float ex, float ey, float* out3);
void     orc_oct_encode(const float* n3, float* out2);
float    orc_sample_blue_noise(int x, int y, int sample_index, int dim, const uint8_t* sobol, const uint8_t* scrambling_ranking);
void     orc_world_position_from_depth(float u, float v, float depth, const float* view_proj_inverse, float* out3);
}
