// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Pinned against the reference's own shaders: tests/test_ref_shaders.py.
// Restatement of /root/reference/src/shaders/reprojection.glsl (whole file) with its three
// compile-time variants expressed as template flags:
//   SINGLE  = REPROJECTION_SINGLE_COLOR_CHANNEL   (shadows, AO)
//   MOMENTS = REPROJECTION_MOMENTS                (shadows, reflections)
//   REFL    = REPROJECTION_REFLECTIONS            (reflections)
#pragma once
#include "orc_common.h"

namespace orc {

#define ORC_NORMAL_DISTANCE 0.1f // reprojection.glsl:6
#define ORC_PLANE_DISTANCE 5.0f  // reprojection.glsl:7

// reprojection.glsl:52-67 (+ the four checks at :11-48)
static inline bool is_reprojection_valid(int cx, int cy, vec3 current_pos, vec3 history_pos, vec3 current_normal, vec3 history_normal, float current_mesh_id, float history_mesh_id, int w, int h)
{
    if (cx < 0 || cy < 0 || cx > w - 1 || cy > h - 1) return false;           // out_of_frame
    if (!(current_mesh_id == history_mesh_id)) return false;                  // mesh id
    vec3  to_current    = current_pos - history_pos;                          // plane distance
    float dist_to_plane = std::fabs(dot(to_current, current_normal));
    if (dist_to_plane > ORC_PLANE_DISTANCE) return false;
    float nd = std::fabs(dot(current_normal, history_normal));                // normals: pow(|d|,2) > 0.1
    if (!(nd * nd > ORC_NORMAL_DISTANCE)) return false;
    return true;
}

// reprojection.glsl:78-96
static inline vec2 virtual_point_reprojection(int cx, int cy, int w, int h, float depth, float ray_length, vec3 cam_pos, const mat4& view_proj_inverse, const mat4& prev_view_proj)
{
    // NB: the shader uses current_coord / size WITHOUT the half-pixel offset here.
    float tu = (float)cx / (float)w, tv = (float)cy / (float)h;
    vec3  ray_origin = world_position_from_depth(tu, tv, depth, view_proj_inverse);
    vec3  camera_ray = ray_origin - cam_pos;
    float camera_ray_length = length(camera_ray);
    camera_ray       = normalize(camera_ray);
    vec3 hp          = cam_pos + camera_ray * (camera_ray_length + ray_length);
    vec4 rp          = mul(prev_view_proj, vec4 { hp.x, hp.y, hp.z, 1.0f });
    float px = rp.x / rp.w, py = rp.y / rp.w;
    return vec2 { (px * 0.5f + 0.5f) * (float)w, (py * 0.5f + 0.5f) * (float)h };
}

struct ReprojectIn
{
    int   x, y;
    float depth;
    // reflections only
    vec3        cam_pos;
    const mat4* prev_view_proj;
    float       ray_length;
    const mat4* view_proj_inverse;
    ImgH<4>     gb2, gb3;            // current, at pass mip
    ImgH<4>     pgb2, pgb3;          // previous
    ImgF        pdepth;
    int         w, h;                // textureSize(history_output)
};

// history colour image: C channels fp16; moments image RGBA16F (rg = moments, b = length);
// length image R16F for the non-moments variant.
template <bool SINGLE, bool MOMENTS, bool REFL, int HC>
static inline bool reproject(const ReprojectIn& in, const ImgH<HC>& hist, const ImgH<4>* hist_moments, const ImgH<1>* hist_length, float* history_color /*1 or 3*/, float* history_moments /*2*/, float* history_length)
{
    const int   w = in.w, h = in.h;
    const float fw = (float)w, fh = (float)h;
    const float tu = ((float)in.x + 0.5f) / fw, tv = ((float)in.y + 0.5f) / fh;

    const float g2x = in.gb2.fetch(in.x, in.y, 0), g2y = in.gb2.fetch(in.x, in.y, 1);
    const float mvx = in.gb2.fetch(in.x, in.y, 2), mvy = in.gb2.fetch(in.x, in.y, 3);
    const vec3  current_normal  = octohedral_to_direction(g2x, g2y);
    const float current_mesh_id = in.gb3.fetch(in.x, in.y, 2);
    const vec3  current_pos     = world_position_from_depth(tu, tv, in.depth, *in.view_proj_inverse);

    int   hcx, hcy;   // history_coord
    float hfx, hfy;   // history_coord_floor
    float htu, htv;   // history_tex_coord
    if (REFL)
    {
        const float curvature = in.gb3.fetch(in.x, in.y, 1);
        htu = tu + mvx; htv = tv + mvy;
        float rx = (float)in.x + mvx * fw, ry = (float)in.y + mvy * fh; // surface_point_reprojection
        if (in.ray_length > 0.0f && curvature == 0.0f)
        {
            vec2 vp = virtual_point_reprojection(in.x, in.y, w, h, in.depth, in.ray_length, in.cam_pos, *in.view_proj_inverse, *in.prev_view_proj);
            rx = vp.x; ry = vp.y;
        }
        hcx = (int)rx; hcy = (int)ry;
        hfx = rx; hfy = ry;
    }
    else
    {
        hcx = (int)(((float)in.x + mvx * fw) + 0.5f);
        hcy = (int)(((float)in.y + mvy * fh) + 0.5f);
        hfx = (float)in.x + mvx * fw;
        hfy = (float)in.y + mvy * fh;
        htu = tu + mvx; htv = tv + mvy;
    }

    const int NC = SINGLE ? 1 : 3;
    for (int c = 0; c < NC; c++) history_color[c] = 0.0f;
    if (MOMENTS) { history_moments[0] = 0.0f; history_moments[1] = 0.0f; }

    bool      v[4];
    const int offx[4] = { 0, 1, 0, 1 }, offy[4] = { 0, 0, 1, 1 };
    const int bx = (int)hfx, by = (int)hfy; // ivec2(vec2): truncation toward zero (quirk 4)

    bool valid = false;
    for (int s = 0; s < 4; s++)
    {
        int   lx = bx + offx[s], ly = by + offy[s];
        float sdepth = in.pdepth.fetch(lx, ly);
        vec3  hn     = octohedral_to_direction(in.pgb2.fetch(lx, ly, 0), in.pgb2.fetch(lx, ly, 1));
        float hmid   = in.pgb3.fetch(lx, ly, 2);
        vec3  hpos   = world_position_from_depth(htu, htv, sdepth, *in.view_proj_inverse);
        v[s]         = is_reprojection_valid(hcx, hcy, current_pos, hpos, current_normal, hn, current_mesh_id, hmid, w, h);
        valid        = valid || v[s];
    }

    if (valid)
    {
        float sumw = 0.0f;
        float fx = fractf(hfx), fy = fractf(hfy);
        float wgt[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };
        for (int c = 0; c < NC; c++) history_color[c] = 0.0f;
        if (MOMENTS) { history_moments[0] = 0.0f; history_moments[1] = 0.0f; }
        for (int s = 0; s < 4; s++)
        {
            int lx = bx + offx[s], ly = by + offy[s];
            if (v[s])
            {
                for (int c = 0; c < NC; c++) history_color[c] += wgt[s] * hist.fetch(lx, ly, c);
                if (MOMENTS)
                {
                    history_moments[0] += wgt[s] * hist_moments->fetch(lx, ly, 0);
                    history_moments[1] += wgt[s] * hist_moments->fetch(lx, ly, 1);
                }
                sumw += wgt[s];
            }
        }
        valid = (sumw >= 0.01f);
        for (int c = 0; c < NC; c++) history_color[c] = valid ? history_color[c] / sumw : 0.0f;
        if (MOMENTS)
        {
            history_moments[0] = valid ? history_moments[0] / sumw : 0.0f;
            history_moments[1] = valid ? history_moments[1] / sumw : 0.0f;
        }
    }
    if (!valid)
    {
        float cnt = 0.0f;
        for (int yy = -1; yy <= 1; yy++)
            for (int xx = -1; xx <= 1; xx++)
            {
                int   px = hcx + xx, py = hcy + yy;
                float sdepth = in.pdepth.fetch(px, py);
                vec3  hn     = octohedral_to_direction(in.pgb2.fetch(px, py, 0), in.pgb2.fetch(px, py, 1));
                float hmid   = in.pgb3.fetch(px, py, 2);
                vec3  hpos   = world_position_from_depth(htu, htv, sdepth, *in.view_proj_inverse);
                if (is_reprojection_valid(hcx, hcy, current_pos, hpos, current_normal, hn, current_mesh_id, hmid, w, h))
                {
                    for (int c = 0; c < NC; c++) history_color[c] += hist.fetch(px, py, c);
                    if (MOMENTS)
                    {
                        history_moments[0] += hist_moments->fetch(px, py, 0);
                        history_moments[1] += hist_moments->fetch(px, py, 1);
                    }
                    cnt += 1.0f;
                }
            }
        if (cnt > 0.0f)
        {
            valid = true;
            for (int c = 0; c < NC; c++) history_color[c] = history_color[c] / cnt;
            if (MOMENTS)
            {
                history_moments[0] = history_moments[0] / cnt;
                history_moments[1] = history_moments[1] / cnt;
            }
        }
    }

    if (valid)
    {
        if (MOMENTS) *history_length = hist_moments->fetch(hcx, hcy, 2);
        else *history_length = hist_length->fetch(hcx, hcy, 0);
    }
    else
    {
        for (int c = 0; c < NC; c++) history_color[c] = 0.0f;
        if (MOMENTS) { history_moments[0] = 0.0f; history_moments[1] = 0.0f; }
        *history_length = 0.0f;
    }
    return valid;
}

} // namespace orc
