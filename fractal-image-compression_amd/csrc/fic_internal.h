// fic_internal.h -- shared by the translation units of the C ABI (fic_capi*.cpp): error reporting, the context type,
// geometry validation, small device-memory helpers and the caches' release hooks.  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fic.h"
#include "fic_device.h"
#include "fic_launch.h"

namespace ficd {

// message + code of the last failure on the calling thread (fic_last_error / fic_last_error_code)
extern thread_local std::string g_err;
extern thread_local int g_err_code;
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return ficd::fail(FIC_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// keeps the calling thread's error across clean-up calls that may overwrite it
struct ErrKeep {
    std::string msg = g_err;
    int code = g_err_code;
    ~ErrKeep() { g_err = msg; g_err_code = code; }
};

// Geometry as the reference derives it (FC:111-116, FC:1019-1022) + what it needs to not throw; out == nullptr: validate only
int make_geometry(int w, int h, int B, int wK, int n_iso, int planes, FicGeom* out);

template <typename T>
int dev_alloc(T** p, size_t count)
{
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return FIC_OK;
}
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// idle single-plane grey contexts of the one-shot / multi-device entries (fic_capi.cpp)
fic_ctx* cache_take(int device, int w, int h, int B, int wK, int n_iso);
void cache_give(fic_ctx* c);
int encode_oneshot(const uint8_t* gray, const int32_t* argb, int w, int h, int B, int wK, int n_iso, int device,
                   int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows);

// what fic_release_cache() frees besides the grey contexts
void release_decoder_arenas();     // fic_capi_decode.cpp
void release_rgb_cache();          // fic_capi_rgb.cpp
void release_comms();              // fic_capi_multi.cpp

}  // namespace ficd

// Device-resident working set of `planes` grey images of one geometry on one device (fic_ctx_* of include/fic.h).
struct fic_ctx {
    int device = 0;
    FicGeom g;
    FicBuffers b;
    FicOutputs o;
    uint8_t* gray_own = nullptr;     // context-owned input copy
    int32_t* argb_stage = nullptr;   // staging for ARGB uploads
    int32_t* collage = nullptr;
    int32_t* host_rec = nullptr;     // pinned host copy of the packed records (fic_ctx_get_results_host)
    uint8_t* decoded = nullptr;      // decoder output image(s)
    FicDecodeState* dec_state = nullptr;   // decoder loop state [planes] and per-pixel squared changes [planes][W*H]
    uint32_t* dec_sq = nullptr;
    void* mfma_poolB = nullptr;      // opt-in matrix-core sweep: B fragments, A fragments, range constants
    void* mfma_rngA = nullptr;
    void* mfma_sw = nullptr;
    int* mfma_rconst = nullptr;
    int mfma_bf16 = 0;               // operand type the fragment stores were built for
    void* q_pool = nullptr;          // k_sweep_q ("sweep" = 6): A fragments, flat-tile flags, B fragments, error bounds, published theta
    void* q_flat = nullptr;
    void* q_rng = nullptr;
    void* q_rngC = nullptr;                  // the sweep columns' isometry copies as bytes (exact evaluation of flagged pairs)
    void* q_E = nullptr;
    void* q_thg = nullptr;
    unsigned int* q_fin = nullptr;           // fused finalise of small launches: finished workgroups per (plane, column group)
    unsigned long long* q_stats = nullptr;   // "sweep_stats" = 1: device counters of k_sweep_q (fic_ctx_sweep_stats)
    uint32_t* d4_rng = nullptr;      // k_sweep_d4: range / domain slots of the group-Fourier form (n_iso = 8, B = 8 / 16)
    uint32_t* d4_pool = nullptr;
    bool have_input = false;
    bool encoded_any = false;
    hipStream_t last_stream = nullptr;
    hipStream_t own_stream = nullptr; // non-blocking stream of the multi-device entry (created on demand)
    int opt_sweep = 0, opt_chunks = 0, opt_time = 0, opt_noflag = 0;
    int last_chunks = 0, last_kind = 0, last_fused = 0, last_tiles_per_chunk = 0;
    std::vector<hipEvent_t> ev;      // pairs start/stop
    double acc_ms = 0.0;
    int acc_n = 0;
    std::mutex mu;
};
