/* gfwarp_testing.h — TEST HOOKS of libgfwarp.  Not part of the operator surface (include/gfwarp.h): nothing a binder of the reference's render loop
 * needs is declared here.  The symbols live in the same shared library so that the GPU tier of the test suite can hold the kernels' own device routines
 * to libm and to their generic twins (tests/test_gpu_math.py), and so that the CPU tier can build the embedded kernel source without a device
 * (tests/test_jit_host.py).
 *
 * Timing ablations (parts of the fused kernel switched OFF for profiling, wrong output by design) are NOT in the library: they exist only in kernel builds that
 * define GFW_TESTING=1 (A/B runs of tools/: GFW_JIT_DEFS="GFW_TESTING=1;GFW_ABLATE_FORCE=<bits>"); gfw_set_option rejects GFW_OPT_KERNEL_VARIANT > 4. */
#ifndef GFWARP_TESTING_H
#define GFWARP_TESTING_H
#include "gfwarp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-side math probes ----
 * gfw_debug_math: out[i] = f(a[i], b[i]) evaluated ON THE DEVICE with the kernels' own routines; host arrays.
 *   op 0 gfw_atanf  1 gfw_tanf  2 gfw_atanf_pos  3 lean a/b  4 generic a/b  5 lean sqrt  6 generic sqrt
 *      7 (float)(i32)`as i32`  8 (float)`as u16`  9 round-half-away  10 (float)`as u8`
 * gfw_debug_selftest: compares a lean routine with its generic twin on `n` device-generated operands
 *   (test 0: divide, operands in the proven range; 1: sqrt; 2: atanf_pos vs atanf over ALL non-negative floats
 *   when n == 0) and returns the number of mismatching results (0 expected), or a negative GFW_ERR_*. */
int   gfw_debug_math(int op, const float *a, const float *b, float *out, size_t n);
long long gfw_debug_selftest(int test, unsigned long long n, unsigned long long seed);

/* Build check of the run-time specialisation path without a device: compiles the kernel source embedded in the library for `arch`
 * ("gfx950") with ';'-separated definitions and a bake header; returns the code object's size in bytes (written to out_path when
 * given), -1 on a compile error (log), -2 when libhiprtc.so is absent. */
long  gfw_debug_jit_compile(const char *arch, const char *defines, const char *bake_header_text, const char *out_path, char *log, size_t cap);

/* Build-time helper of the shipped kernel cache (tools/build_jit_cache.py): the definition list (';'-separated), bake header and cache file name the library
 * would specialise a frame of these planes to — derived exactly as on a device, without one.  `matrices_on_device` as GFW_OPT_MATRICES_ON_DEVICE would be. */
int   gfw_debug_jit_key(int nplanes, const gfw_buffers *planes, const gfw_kernel_params *params, const int *pixel_types, int distortion_model, int digital_lens,
                        const float *host_matrices, int matrix_count, int matrices_on_device, const char *arch,
                        char *defs_out, size_t defs_cap, char *header_out, size_t header_cap, char *name_out, size_t name_cap);

/* Identity of the fused kernel's source this library was built from (length and 128-bit hash of the text it embeds for run-time specialisation; the ahead-of-time
 * kernels are compiled from the same files): measurements stored beside the repository (profiles/ *_traffic.json) name the source they were taken on, and bench.py
 * quotes them only for the library that matches.  Returns the string's length, or a negative GFW_ERR_*. */
int   gfw_debug_source_id(char *out, size_t cap);

/* The host side of the certified first pass of a radial lens model other than the fisheye (round 6: GFW_MODEL_GOPRO), without a device: `table` receives the
 * 8193 float pairs (T(r_i), T(r_i+1) - T(r_i)) over r in [0, r_max] (or NULL), out7 = {r_max, max T, max |T'|, bound on |T''|, table error, noise of the exact
 * path's Newton result, min POLY'}.  1: certificate derived; 0: the host declines for these coefficients / this range (the clip keeps the exact first pass). */
int   gfw_debug_p1_radial(const gfw_kernel_params *params, int distortion_model, double r_max, float *table, double *out7);

/* How many launches of the per-plane kernel on this context served TWO planes (round 6: EWA on the U and V planes of a planar frame — one set of coordinates and
 * tap weights, two sums; the backend name stays "plane_generic").  -1 for a null context. */
long long gfw_debug_paired_launches(gfw_ctx *ctx);

/* How many frames one launch of a gfw_undistort_clip call takes (no device needed): frames of `bytes_per_frame` of source + destination under a budget of
 * `budget_bytes` per launch (0: the library's, 1.1 GB or GFW_CLIP_LAUNCH_MB), `frames_in_call` frames dealt evenly over the launches that needs (0: frames held
 * from per-plane calls — the cap alone).  Between 2 and GFW_CLIP_FRAMES_MAX. */
int   gfw_debug_frames_per_launch(unsigned long long bytes_per_frame, unsigned long long budget_bytes, int frames_in_call);

#ifdef __cplusplus
}
#endif
#endif /* GFWARP_TESTING_H */
