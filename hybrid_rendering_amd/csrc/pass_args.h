// Kernel argument blocks of the denoise / resolve stages, shared by the exact kernels (shadows.hip, ao.hip, reflections.hip,
// ddgi.hip) and their tolerance-mode counterparts (denoise_fast.hip): both read the same images and write the same outputs.
#pragma once
#include "reproject.h"
#include "shading.h"
#include "upsample.h"

namespace hr {

// A sort of last launch's tile costs into the trace kernel's next launch order, riding along as extra grid rows of a temporal kernel
// (tile_order.h).  groups == 0: none.
struct TileSortArgs
{
    const uint16_t* cost;
    uint32_t*       order;
    int             n, groups;
    int             row0;   // first grid row of the sort's workgroups
    int             min_spread;   // TileOrder::min_spread
};

// Row bands, tolerance mode: the geometry records of the rows a band reads HISTORY from but does not compute — the part of
// hr_band.history_halo beyond hr_band.halo.  A record is a copy of two words of the CURRENT G-buffer texel, which the caller keeps readable
// on those rows (hr_api.h hr_band), so a few extra workgroups of the temporal launch (grid rows >= row0) write them: rows [a0, a1) above and
// [b0, b1) below the computed rows.  row0 < 0: none.
struct GeoApronArgs
{
    const void* gb2;
    const void* gb3;
    void*       out;      // this frame's records {GB2.x (oct normal), GB3.y (mesh id | linear z)}
    int         w, a0, a1, b0, b1;
    int         row0;
};

struct TemporalArgs
{
    float           vpi[16];
    const uint32_t* mask;
    int             mw, mh;         // mask image dims (full frame)
    ImgRGBA16F      gb2, gb3, pgb2, pgb3;
    ImgR32F         depth, pdepth;
    ImgRG16F        hist;           // previous à-trous feedback image (vis, var)
    ImgRGBA16F      hist_moments;
    uint32_t*       out;            // RG16F
    uint2*          out_moments;    // RGBA16F
    uint8_t*        tile_class;
    float4*         nd;             // decoded normal.xyz + linear z (GB3.w), written for the a-trous iterations
    const void*     geo_hist;       // tolerance mode: LAST frame's `nd` records {oct normal, mesh id | linear z} when they stand for the
                                    // caller's previous G-buffer (hr_shadows_temporal decides), else nullptr -> pgb2 / pgb3 are read
    int             w, h, y0, y1;
    int             tiles_x, tiles_y, tile_y0;
    float           alpha, moments_alpha;
    int             debug_skip_reproject; // developer ablation switch (HR_DEBUG_SKIP_REPROJECT)
    uint32_t*       apron_flag;     // row bands only (else nullptr): set to 1 when a history tap of a pixel of the band proper [band_y0,
                                    // band_y1) fell on a row of the image that is not resident on this GPU — the motion exceeded
                                    // hr_band.history_halo (the tap read as disoccluded).  Halo rows are the neighbour's to get right.
    int             band_y0, band_y1;
    TileSortArgs    sort;           // rides along as extra grid rows (tolerance-mode kernel only)
    GeoApronArgs    apron;          // rides along behind the sort (tolerance-mode kernel, row bands only)
};

struct AtrousArgs
{
    ImgRG16F       in;
    const float4*  nd;      // decoded normal + linear z of every pixel (k_shadows_temporal)
    const uint8_t* tile_class;
    uint32_t*      out;
    uint32_t*      out2;    // feedback copy (prev_image) or nullptr
    int            w, h, y0, y1, tiles_x;
    int            radius, step;
    float          phi_visibility, phi_normal, sigma_depth, power;
};

struct AOTemporalArgs
{
    float           vpi[16];
    const uint32_t* mask;
    int             mw, mh, spp;
    ImgRGBA16F      gb2, gb3, pgb2, pgb3;
    ImgR32F         depth, pdepth;
    ImgR16F         hist, hist_len;
    uint16_t*       out;
    uint16_t*       out_len;
    uint8_t*        tile_class;
    const void*     geo_hist;       // tolerance mode: last frame's records {oct normal, mesh id | AO} when they stand for the caller's
                                    // previous G-buffer + the AO history (hr_ao_temporal decides), else nullptr
    void*           geo_out;        // tolerance mode: this frame's records, or nullptr
    int             geo_band;       // 1: row band — only the geometry half of the records is read, the AO history comes from `hist` (hr_ao_temporal)
    int             w, h, y0, y1;
    int             tiles_x, tiles_y, tile_y0;
    float           alpha;
    uint32_t*       apron_flag;     // see TemporalArgs
    int             band_y0, band_y1;
    TileSortArgs    sort;           // rides along as extra grid rows (tolerance-mode kernel only)
};

struct AOBlurArgs
{
    ImgR16F        in;
    ImgR32F        depth;
    ImgRGBA16F     gb2;
    const uint8_t* tile_class;
    uint16_t*      out;
    float          zbp[4];
    int            w, h, y0, y1, tiles_x;
    int            dx, dy, radius;
};

struct ReflTemporalArgs
{
    float       vpi[16], pvp[16];
    float       cam[3];
    ImgRGBA16F  in, gb2, gb3, pgb2, pgb3, hist, hist_moments;
    ImgR32F     depth, pdepth;
    uint2*      out;
    uint2*      out_moments;
    uint8_t*    tile_class;
    const void* geo_hist;           // tolerance mode: last frame's records {oct normal, mesh id | linear z} when they stand for the caller's
                                    // previous G-buffer (hr_reflections_temporal decides), else nullptr -> pgb2 / pgb3 are read
    void*       geo_out;            // tolerance mode: this frame's records, or nullptr
    int         w, h, y0, y1, tiles_x;
    float       alpha, moments_alpha;
    int         approximate_with_ddgi, moving;
    uint32_t*   apron_flag;         // see TemporalArgs
    int         band_y0, band_y1;
    TileSortArgs sort;              // rides along as extra grid rows (tolerance-mode kernel only)
};

struct ReflAtrousArgs
{
    ImgRGBA16F     in, gb2, gb3;
    ImgR32F        depth;
    const void*    geo;     // tolerance mode: this frame's records {oct normal, mesh id | linear z} (the taps' two half-used G-buffer
                            // gathers become one 8-byte load), or nullptr
    const uint8_t* tile_class;
    uint2*         out;
    uint2*         out2;
    int            w, h, y0, y1, tiles_x, radius, step;
    float          phi_color, phi_normal, sigma_depth;
    int            approximate_with_ddgi;
};

struct DDGISampleArgs
{
    DDGIU        d;
    float        vpi[16];
    float        cam[3];
    const float* depth;
    const uint2* gb2;
    AtlasRGBA    irr;
    AtlasRG      dep;
    uint2*       out;
    int          w, h, y0, y1;   // rows [y0, y1) of the image are produced (row band)
    float        gi_intensity;
};

// ---- tolerance-mode ("fast") launchers, denoise_fast.hip ----------------------------------------------------------------
// hr_*_params.exact == 0 selects them: same inputs, same outputs, same tile classification rule; fp32 arithmetic through the
// hardware's v_rcp / v_rsq / v_sqrt / v_exp / v_log, contracted FMAs and re-associated sums.  The visibility
// masks never pass through them (the trace kernels have one mode).
void launch_shadows_temporal_fast(const TemporalArgs& a, int n_tiles, hipStream_t st);
void launch_shadows_atrous_fast(const AtrousArgs& a, hipStream_t st);
bool launch_shadows_atrous01_fast(const AtrousArgs& a, uint32_t* out_first2, float power1, hipStream_t st);   // iterations 0 + 1 in one launch
void launch_ao_temporal_fast(const AOTemporalArgs& a, int n_tiles, hipStream_t st);
void launch_ao_blur_fast(const AOBlurArgs& a, hipStream_t st);
bool launch_ao_blur_xy_fast(const AOBlurArgs& a, hipStream_t st);   // X + Y in one launch (radius 4); false: not available for these arguments
void launch_refl_temporal_fast(const ReflTemporalArgs& a, hipStream_t st);
void launch_refl_atrous_fast(const ReflAtrousArgs& a, hipStream_t st);
bool launch_refl_atrous01_fast(const ReflAtrousArgs& a, uint2* out_first2, hipStream_t st);   // iterations 0 + 1 in one launch
void launch_ddgi_sample_fast(const DDGISampleArgs& a, hipStream_t st);
void launch_upsample_fast(const UpsampleArgs& a, hipStream_t st);

} // namespace hr
