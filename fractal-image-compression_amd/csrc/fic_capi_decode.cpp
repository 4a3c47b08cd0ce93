// fic_capi_decode.cpp -- C ABI, decoder entries: FractalCompression.decode on .run streams (FC:547-553 -> decodeGreyScale
// FC:356-421, decodeRGB FC:430-508) and the decode of a context's own codebook.  Host-side orchestration only.
#include "fic_internal.h"

using namespace ficd;

// ---- decoder (decodeGreyScale FC:356-421) ------------------------------------------------------
namespace {
// Device arenas of the stream decoders, kept between calls (the GUI decodes after every encode, CTL:178-179): one
// allocation per (device, size class) instead of four hipMalloc/hipFree per call.  fic_release_cache() frees them.
struct Arena {
    int device = -1;
    size_t bytes = 0;
    char* base = nullptr;
};
std::mutex g_arena_mu;
std::vector<Arena> g_arenas;
constexpr size_t kArenaSlots = 4;

int arena_take(int device, size_t bytes, Arena* out)
{
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        for (size_t i = g_arenas.size(); i-- > 0;)
            if (g_arenas[i].device == device && g_arenas[i].bytes >= bytes && g_arenas[i].bytes <= 2 * bytes + (1u << 20)) {
                *out = g_arenas[i];
                g_arenas.erase(g_arenas.begin() + (long)i);
                return FIC_OK;
            }
    }
    out->device = device;
    out->bytes = bytes;
    HIP_TRY(hipMalloc((void**)&out->base, bytes));
    return FIC_OK;
}
void arena_give(const Arena& a)
{
    Arena evict;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        g_arenas.push_back(a);
        if (g_arenas.size() <= kArenaSlots) return;
        evict = g_arenas.front();
        g_arenas.erase(g_arenas.begin());
    }
    (void)hipSetDevice(evict.device);
    (void)hipFree(evict.base);
}
}  // namespace

void ficd::release_decoder_arenas()
{
    std::vector<Arena> drop;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        drop.swap(g_arenas);
    }
    for (const Arena& a : drop) { (void)hipSetDevice(a.device); (void)hipFree(a.base); }
}

static thread_local int g_last_sum_fallbacks = 0;

extern "C" {

// Runs the reconstruction loop on the device.  The first 8 iterations are enqueued in one go, later ones in pairs, and the
// per-plane loop state is read back after each group (a converging decode takes 6-9 iterations): one host sync per
// group, none per iteration; iterations enqueued behind the last one exit at once.
//   d_state [planes], d_sqbuf u32 [planes][W*H]: scratch of the caller
static int run_decode_loop(const FicGeom& g, uint8_t* d_scaled, uint8_t* d_image, const int32_t* d_qrows,
                           const int32_t* d_iso, FicDecodeState* d_state, uint32_t* d_sqbuf, const float* avg_in,
                           float* avg_out, int* iters_out, int* seq_out, hipStream_t s)
{
    const size_t P = (size_t)g.planes;
    std::vector<FicDecodeState> st(P);
    memset(st.data(), 0, P * sizeof(FicDecodeState));
    for (size_t p = 0; p < P; p++) st[p].avg = avg_in ? avg_in[p] : 0.0f;   // static avgError is never reset (FC:20)
    int rc = FIC_OK;
    hipError_t e = hipMemcpyAsync(d_state, st.data(), P * sizeof(FicDecodeState), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_image, 128, P * g.W * g.H, s);        // generateGrayImage FC:1142-1148
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "decode init: %s", hipGetErrorString(e));
    for (int counter = 0; rc == FIC_OK && counter < 50; counter++) {
        if (fic_launch_decode_iteration(d_scaled, d_image, d_qrows, d_iso, d_state, d_sqbuf, counter, g, s)) {
            rc = fail(FIC_E_HIP, "decode iteration launch failed");
            break;
        }
        // a converging decode takes 6-9 iterations: look at the loop state after 8, then after every second iteration
        if (counter == 7 || (counter > 7 && (counter & 1)) || counter == 49) {
            e = hipMemcpyAsync(st.data(), d_state, P * sizeof(FicDecodeState), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) { rc = fail(FIC_E_HIP, "decode readback: %s", hipGetErrorString(e)); break; }
            bool all = true;
            for (size_t p = 0; p < P; p++) all = all && st[p].done;
            if (all) break;
        }
    }
    if (rc != FIC_OK) return rc;
    for (size_t p = 0; p < P; p++) {
        if (st[p].bad_index)
            return fail(FIC_E_ARGUMENT, "decode: a codebook row of plane %zu points outside the domain pool "
                                        "(ArrayIndexOutOfBounds at FC:394 in the reference)", p);
        if (avg_out) avg_out[p] = st[p].avg_out;
        if (iters_out) iters_out[p] = st[p].iters;
        if (seq_out) seq_out[p] = st[p].seq_sums;
    }
    return FIC_OK;
}

int fic_ctx_decode_host(fic_ctx* c, uint8_t* gray_out, float* avg_error_out, int* iterations_out)
{
    if (!c || !gray_out) return fail(FIC_E_ARGUMENT, "fic_ctx_decode_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_ctx_decode_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    const FicGeom& g = c->g;
    size_t npix = (size_t)g.planes * g.W * g.H;
    if (!c->decoded) { int rc = dev_alloc(&c->decoded, npix); if (rc) return rc; }
    if (!c->dec_state) { int rc = dev_alloc(&c->dec_state, (size_t)g.planes); if (rc) return rc; }
    if (!c->dec_sq) { int rc = dev_alloc(&c->dec_sq, fic_decode_sq_words((size_t)g.planes, (size_t)g.W * g.H)); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    int rc = run_decode_loop(g, c->b.scaled, c->decoded, c->o.qrows, g.n_iso > 1 ? c->o.iso : nullptr, c->dec_state, c->dec_sq,
                             nullptr, avg_error_out, iterations_out, nullptr, c->last_stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(gray_out, c->decoded, npix, hipMemcpyDeviceToHost));
    return FIC_OK;
}

static int32_t run_be32(const uint8_t* run, int64_t off)
{
    return (int32_t)(((uint32_t)run[off] << 24) | ((uint32_t)run[off + 1] << 16) | ((uint32_t)run[off + 2] << 8) |
                     (uint32_t)run[off + 3]);
}

static int decode_gray_run_impl(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity, int* w_out,
                                int* h_out, float* avg_error_io, int* iterations, int* seq_sums)
{
    if (!run || len < 20) return fail(FIC_E_ARGUMENT, "fic_decode_gray_run: stream shorter than the 20-byte header");
    if (run_be32(run, 0) != 0)
        return fail(FIC_E_NOT_GREY, "fic_decode_gray_run: isRGB = %d (FC:548-552 dispatches to decodeRGB)", run_be32(run, 0));
    const int w = run_be32(run, 4), h = run_be32(run, 8), B = run_be32(run, 12), wK = run_be32(run, 16);
    FicGeom g;
    int rc = make_geometry(w, h, B, wK, 1, 1, &g);
    if (rc) return rc;
    if (w_out) *w_out = w;
    if (h_out) *h_out = h;
    if (len < 20 + 12 * (int64_t)g.Nr)
        return fail(FIC_E_ARGUMENT, "fic_decode_gray_run: %lld bytes, need %lld (EOFException in the reference)",
                    (long long)len, (long long)(20 + 12 * (int64_t)g.Nr));
    if (!gray_out || capacity < (int64_t)w * h) return fail(FIC_E_CAPACITY, "fic_decode_gray_run: output needs %d bytes", w * h);
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d (this library has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    std::vector<int32_t> q((size_t)g.Nr * 3);
    for (size_t i = 0; i < q.size(); i++) q[i] = run_be32(run, 20 + 4 * (int64_t)i);          // FC:372-374
    const size_t npix = (size_t)w * h;
    const size_t o_scaled = 0, o_image = o_scaled + align256((size_t)g.Ws * g.Hs), o_q = o_image + align256(npix),
                 o_state = o_q + align256(q.size() * 4), o_sq = o_state + align256(sizeof(FicDecodeState)),
                 total = o_sq + align256(fic_decode_sq_words(1, npix) * 4);
    Arena ar;
    rc = arena_take(device, total, &ar);
    if (rc) return rc;
    hipError_t e = hipMemcpy(ar.base + o_q, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_gray_run: %s", hipGetErrorString(e));
    float avg = avg_error_io ? *avg_error_io : 0.0f;
    if (rc == FIC_OK)
        rc = run_decode_loop(g, (uint8_t*)(ar.base + o_scaled), (uint8_t*)(ar.base + o_image), (const int32_t*)(ar.base + o_q), nullptr,
                             (FicDecodeState*)(ar.base + o_state), (uint32_t*)(ar.base + o_sq), &avg, &avg, iterations, seq_sums, nullptr);
    if (rc == FIC_OK) {
        e = hipMemcpy(gray_out, ar.base + o_image, npix, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_gray_run: %s", hipGetErrorString(e));
    }
    if (rc == FIC_OK && avg_error_io) *avg_error_io = avg;
    arena_give(ar);
    return rc;
}

int fic_decode_gray_run(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity, int* w_out,
                        int* h_out, float* avg_error_io, int* iterations)
{
    return decode_gray_run_impl(run, len, device, gray_out, capacity, w_out, h_out, avg_error_io, iterations, nullptr);
}

// Test hook: the decoder's reproduction of Java's `avgError += (float) v[i]` loop (FC:407) on arbitrary values.
int fic_debug_float_sum(int device, float carry, const uint32_t* vals, int count, float* out)
{
    if (!vals || !out || count < 0) return fail(FIC_E_ARGUMENT, "fic_debug_float_sum: bad argument");
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    uint32_t* d = nullptr;
    float* r = nullptr;
    uint32_t* maps = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)(count + 4) * 4));
    hipError_t e = hipMalloc((void**)&r, 8);
    if (e == hipSuccess) e = hipMalloc((void**)&maps, (fic_float_sum_map_words((size_t)count) + 4) * 4);
    if (e == hipSuccess) e = hipMemcpy(d, vals, (size_t)count * 4, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? FIC_OK : fail(FIC_E_HIP, "fic_debug_float_sum: %s", hipGetErrorString(e));
    if (rc == FIC_OK && fic_launch_float_sum_probe(carry, d, count, maps, r, nullptr)) rc = fail(FIC_E_HIP, "k_float_sum_probe launch failed");
    if (rc == FIC_OK) {
        float two[2] = {0.0f, 0.0f};
        e = hipMemcpy(two, r, 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_debug_float_sum: %s", hipGetErrorString(e));
        out[0] = two[0];
        g_last_sum_fallbacks = (int)two[1];
    }
    (void)hipFree(d);
    if (r) (void)hipFree(r);
    if (maps) (void)hipFree(maps);
    return rc;
}

// Test hook: segments of the last fic_debug_float_sum on this thread that went through the sequential-order path.
int fic_debug_float_sum_fallbacks(void) { return g_last_sum_fallbacks; }

// Test hook: fic_decode_gray_run that also reports how many iterations needed the sequential (Java-order) float sum.
int fic_debug_decode_gray_run(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity,
                              float* avg_error_io, int* iterations, int* seq_sums)
{
    return decode_gray_run_impl(run, len, device, gray_out, capacity, nullptr, nullptr, avg_error_io, iterations, seq_sums);
}

// ---- decodeRGB (FC:430-508) -----------------------------------------------------------------------
int fic_decode_rgb_run(const uint8_t* run, int64_t len, int device, int32_t* argb_out, int64_t capacity_pixels,
                       int* w_out, int* h_out, float* avg_error_io, int* iterations)
{
    if (!run || len < 20) return fail(FIC_E_ARGUMENT, "fic_decode_rgb_run: stream shorter than the 20-byte header");
    if (run_be32(run, 0) == 0) return fail(FIC_E_ARGUMENT, "fic_decode_rgb_run: isRGB = 0 (FC:548-550 dispatches to decodeGreyScale)");
    const int w = run_be32(run, 4), h = run_be32(run, 8), B = run_be32(run, 12), wK = run_be32(run, 16);
    FicGeom g;
    int rc = make_geometry(w, h, B, wK, 1, 1, &g);
    if (rc) return rc;
    if (w_out) *w_out = w;
    if (h_out) *h_out = h;
    if (len < 20 + 20 * (int64_t)g.Nr)
        return fail(FIC_E_ARGUMENT, "fic_decode_rgb_run: %lld bytes, need %lld (EOFException in the reference)",
                    (long long)len, (long long)(20 + 20 * (int64_t)g.Nr));
    if (!argb_out || capacity_pixels < (int64_t)w * h) return fail(FIC_E_CAPACITY, "fic_decode_rgb_run: output needs %d ints", w * h);
    int ndev = fic_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return fail(FIC_E_NO_DEVICE, "no HIP device %d (this library has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    std::vector<int32_t> q((size_t)g.Nr * 5);
    for (size_t i = 0; i < q.size(); i++) q[i] = run_be32(run, 20 + 4 * (int64_t)i);          // FC:446-450
    const size_t npix = (size_t)w * h;
    std::vector<int32_t> init(npix, (int32_t)0xff808080u);                          // generateGrayImage FC:1142-1148
    const size_t o_scaled = 0, o_image = o_scaled + align256((size_t)g.Ws * g.Hs * 4), o_q = o_image + align256(npix * 4),
                 o_state = o_q + align256(q.size() * 4), o_sq = o_state + align256(sizeof(FicDecodeState)),
                 total = o_sq + align256(fic_decode_sq_words(1, npix) * 4);
    Arena ar;
    rc = arena_take(device, total, &ar);
    if (rc) return rc;
    int32_t* d_scaled = (int32_t*)(ar.base + o_scaled);
    int32_t* d_image = (int32_t*)(ar.base + o_image);
    int32_t* d_q = (int32_t*)(ar.base + o_q);
    FicDecodeState* d_state = (FicDecodeState*)(ar.base + o_state);
    uint32_t* d_sq = (uint32_t*)(ar.base + o_sq);
    FicDecodeState st;
    memset(&st, 0, sizeof(st));
    st.avg = avg_error_io ? *avg_error_io : 0.0f;
    hipError_t e = hipMemcpy(d_q, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_image, init.data(), npix * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_state, &st, sizeof(st), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_rgb_run: %s", hipGetErrorString(e));
    for (int counter = 0; rc == FIC_OK && counter < 50; counter++) {
        if (fic_launch_decode_iteration_rgb(d_scaled, d_image, d_q, d_state, d_sq, counter, g, nullptr)) {
            rc = fail(FIC_E_HIP, "decodeRGB iteration launch failed");
            break;
        }
        if ((counter & 7) == 7 || counter == 49) {
            e = hipMemcpy(&st, d_state, sizeof(st), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = fail(FIC_E_HIP, "decodeRGB readback: %s", hipGetErrorString(e)); break; }
            if (st.done) break;
        }
    }
    if (rc == FIC_OK && st.bad_index)
        rc = fail(FIC_E_ARGUMENT, "decodeRGB: a codebook row points outside the domain pool (ArrayIndexOutOfBounds at FC:477)");
    if (rc == FIC_OK) {
        e = hipMemcpy(argb_out, d_image, npix * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "fic_decode_rgb_run: %s", hipGetErrorString(e));
    }
    if (rc == FIC_OK) {
        if (avg_error_io) *avg_error_io = st.avg_out;
        if (iterations) *iterations = st.iters;
    }
    arena_give(ar);
    return rc;
}

}  // extern "C"
