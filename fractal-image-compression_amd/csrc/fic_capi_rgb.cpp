// fic_capi_rgb.cpp -- C ABI, joint-RGB path: encodeRGB (FC:171-219) as one-shot entry and as batched device-resident
// contexts (fic_rgb_ctx_*), decodeRGB from a context's codebook, the RGB writeData branch.  Host-side orchestration only.
#include "fic_internal.h"

using namespace ficd;

extern "C" {

// ---- joint-RGB encode (encodeRGB FC:171-219) -----------------------------------------------------
// A context owns the device working set of `planes` colour images of one geometry (config-5 style batches; the one-shot
// entry keeps a few single-image contexts).  The kernels take the image from their grid: a batch is one launch per stage on the
// caller's stream (only the matrix-core full search runs image by image).  The covariance sums stay sequential f32 in the reference's order (FC:781-792: they exceed 2^24).
struct fic_rgb_ctx {
    int device = 0;
    FicGeom g;
    int32_t* argb_own = nullptr;     // context-owned input copy
    const int32_t* argb = nullptr;   // input in use (own copy or the caller's device pointer)
    int32_t* scaled = nullptr;       // per plane: [H/2][W/2]
    uint16_t* pool_sum = nullptr;    // [N_d][n]
    float* pool_cf = nullptr;        // [N_d][n] (full search at B = 4 / 8)
    FicRgbDomStat* pool_st = nullptr;
    int16_t* rng_t = nullptr;
    FicRgbRngStat* rng_st = nullptr;
    unsigned long long* key = nullptr;
    int32_t *idx_local = nullptr, *idx_global = nullptr, *qrows = nullptr, *collage = nullptr;
    float *a = nullptr, *bR = nullptr, *bG = nullptr, *bB = nullptr;
    int32_t* dec_image = nullptr;    // decoder: image, scaled image, state, per-pixel squared changes (one plane at a time)
    int32_t* dec_scaled = nullptr;
    FicDecodeState* dec_state = nullptr;
    uint32_t* dec_sq = nullptr;
    FicRgbQ q;                       // matrix-core full search (allocated on first use; one image at a time, stream-ordered)
    int opt_sweep = 0;               // 0 auto, 1 VALU sweeps, 2 matrix-core full search
    int opt_chunks = 0;              // pool chunks of the matrix-core sweep (0 = automatic)
    int last_sweep = 0;              // what the last encode ran: 1 / 2
    bool have_input = false, encoded_any = false, have_collage = false;
    hipStream_t last_stream = nullptr;
    std::mutex mu;
};

namespace {
void rgb_free_all(fic_rgb_ctx* c)
{
    (void)hipSetDevice(c->device);
    void* ptrs[] = {c->argb_own, c->scaled, c->pool_sum, c->pool_cf, c->pool_st, c->rng_t, c->rng_st, c->key, c->idx_local,
                    c->idx_global, c->qrows, c->collage, c->a, c->bR, c->bG, c->bB, c->dec_image, c->dec_scaled, c->dec_state, c->dec_sq,
                    c->q.poolQ, c->q.dflat, c->q.rngQ, c->q.qst, c->q.rngE, c->q.theta_g, c->q.amax};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
}
// buffers of plane p as the per-image kernels expect them
void rgb_plane(const fic_rgb_ctx* c, int p, FicRgbBuffers* b, FicRgbOutputs* o)
{
    const FicGeom& g = c->g;
    const size_t P = (size_t)p, nd = (size_t)g.Nd, nr = (size_t)g.Nr, n = (size_t)g.n;
    b->argb = const_cast<int32_t*>(c->argb) + P * g.W * g.H;
    b->scaled = c->scaled + P * g.Ws * g.Hs;
    b->pool_sum = c->pool_sum + P * nd * n;
    b->pool_cf = c->pool_cf ? c->pool_cf + P * nd * n : nullptr;
    b->pool_st = c->pool_st + P * nd;
    b->rng_t = c->rng_t + P * nr * n;
    b->rng_st = c->rng_st + P * nr;
    b->key = c->key + P * nr;
    o->idx_local = c->idx_local + P * nr;
    o->idx_global = c->idx_global + P * nr;
    o->a = c->a + P * nr;
    o->bR = c->bR + P * nr;
    o->bG = c->bG + P * nr;
    o->bB = c->bB + P * nr;
    o->qrows = c->qrows + P * nr * 5;
}
// pool chunks of the matrix-core full search: the policy of the grey q sweep (fic_capi.cpp, q_sweep), or the "chunks" option
void rgb_q_chunks(fic_rgb_ctx* c)
{
    const FicGeom& g = c->g;
    FicRgbQ& q = c->q;
    const int unroll = fic_q_unroll(g.B, 1), CT = fic_q_ct(g.B);
    const int nct = (g.Nr + 31) / 32;
    long long nc = c->opt_chunks;
    if (nc <= 0) {
        const long long base_wg = (nct + CT - 1) / CT;
        const long long resident = 256LL * fic_q_resident(g.B);
        auto ceil_div = [](long long a, long long b) { return (a + b - 1) / b; };
        long long fill = base_wg < resident ? ceil_div(2 * resident, base_wg) : 1;
        if (fill > q.ndtiles / 16) fill = q.ndtiles / 16;
        long long bal = ceil_div(8 * resident, base_wg);
        if (bal > q.ndtiles / 1024) bal = q.ndtiles / 1024;
        nc = fill > bal ? fill : bal;
    }
    if (nc < 1) nc = 1;
    if (nc > q.ndtiles) nc = q.ndtiles;
    int tpc = (int)((q.ndtiles + nc - 1) / nc);
    tpc = (tpc + unroll - 1) / unroll * unroll;
    q.tiles_per_chunk = tpc;
    q.nchunks = (q.ndtiles + tpc - 1) / tpc;
}
// buffers of the matrix-core full search (same shapes as the grey q sweep)
int rgb_q_setup(fic_rgb_ctx* c)
{
    const FicGeom& g = c->g;
    FicRgbQ& q = c->q;
    if (q.poolQ) return FIC_OK;
    const int unroll = fic_q_unroll(g.B, 1), CT = fic_q_ct(g.B);
    const size_t NK = (size_t)g.n / 16;
    q.ndtiles = (g.Nd + 31) / 32;
    q.ndtiles_alloc = q.ndtiles + 2 * unroll;
    const int nct = (g.Nr + 31) / 32;
    q.nct_alloc = ((nct + CT - 1) / CT * CT + CT + 1) & ~1;
    int rc = FIC_OK;
    auto A = [&](hipError_t e) { if (rc == FIC_OK && e != hipSuccess) rc = fail(FIC_E_HIP, "RGB matrix-core buffers: %s", hipGetErrorString(e)); };
    A(hipMalloc(&q.poolQ, (size_t)q.ndtiles_alloc * NK * 64 * 16));
    A(hipMalloc(&q.dflat, (size_t)q.ndtiles_alloc * sizeof(uint32_t)));
    A(hipMalloc(&q.rngQ, (size_t)q.nct_alloc * NK * 64 * 16));
    A(hipMalloc(&q.qst, (size_t)g.Nr * sizeof(FicRngStat)));
    A(hipMalloc(&q.rngE, (size_t)g.Nr * sizeof(float)));
    A(hipMalloc(&q.theta_g, (size_t)q.nct_alloc * 32 * sizeof(uint32_t)));     // per column incl. the padded column tiles (the sweep's fast path reads them)
    A(hipMalloc(&q.amax, 256));
    if (rc != FIC_OK) {
        void* ptrs[] = {q.poolQ, q.dflat, q.rngQ, q.qst, q.rngE, q.theta_g, q.amax};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        q = FicRgbQ();
    }
    return rc;
}
// idle single-image contexts of the one-shot RGB entry, most recently used last
std::mutex g_rgb_mu;
std::vector<fic_rgb_ctx*> g_rgb_cache;
}  // namespace

fic_rgb_ctx* fic_rgb_ctx_create(int device, int w, int h, int B, int wK, int planes)
{
    FicGeom g;
    if (make_geometry(w, h, B, wK, 1, planes, &g)) return nullptr;
    int ndev = fic_device_count();
    if (ndev <= 0) { fail(FIC_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)"); return nullptr; }
    if (device < 0 || device >= ndev) { fail(FIC_E_NO_DEVICE, "device %d out of range (0..%d)", device, ndev - 1); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(FIC_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
    fic_rgb_ctx* c = new fic_rgb_ctx();
    c->device = device;
    c->g = g;
    const size_t P = (size_t)planes, npix = (size_t)w * h, nr = (size_t)g.Nr, nd = (size_t)g.Nd, n = (size_t)g.n;
    int rc = FIC_OK;
    auto A = [&](int r) { if (rc == FIC_OK) rc = r; };
    A(dev_alloc(&c->scaled, P * g.Ws * g.Hs));
    A(dev_alloc(&c->pool_sum, P * nd * n));
    A(dev_alloc(&c->pool_st, P * nd));
    A(dev_alloc(&c->rng_t, P * nr * n));
    A(dev_alloc(&c->rng_st, P * nr));
    A(dev_alloc(&c->key, P * nr));
    A(dev_alloc(&c->idx_local, P * nr));
    A(dev_alloc(&c->idx_global, P * nr));
    A(dev_alloc(&c->a, P * nr));
    A(dev_alloc(&c->bR, P * nr));
    A(dev_alloc(&c->bG, P * nr));
    A(dev_alloc(&c->bB, P * nr));
    A(dev_alloc(&c->qrows, P * nr * 5));
    A(dev_alloc(&c->collage, P * npix));
    if (rc != FIC_OK) {
        rgb_free_all(c);
        delete c;
        return nullptr;
    }
    return c;
}

void fic_rgb_ctx_destroy(fic_rgb_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    rgb_free_all(c);
    delete c;
}

int fic_rgb_ctx_set_argb_host(fic_rgb_ctx* c, const int32_t* argb)
{
    if (!c || !argb) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_argb_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)c->g.planes * c->g.W * c->g.H;
    if (!c->argb_own) { int rc = dev_alloc(&c->argb_own, npix); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->last_stream));     // a previous encode may still read the copy
    HIP_TRY(hipMemcpy(c->argb_own, argb, npix * sizeof(int32_t), hipMemcpyHostToDevice));
    c->argb = c->argb_own;
    c->have_input = true;
    return FIC_OK;
}

int fic_rgb_ctx_set_argb_device(fic_rgb_ctx* c, const void* dev_argb)
{
    if (!c || !dev_argb) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_argb_device: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    c->argb = (const int32_t*)dev_argb;
    c->have_input = true;
    return FIC_OK;
}

int fic_rgb_ctx_encode(fic_rgb_ctx* c, int with_collage, void* hip_stream)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_encode: null context");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->have_input) return fail(FIC_E_STATE, "fic_rgb_ctx_encode: no input image set");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)hip_stream;
    c->last_stream = s;
    const FicGeom& g = c->g;
    // Full search: the matrix-core sweep where it pays (its prep costs ~50 us; the VALU sweep does ~1e11 pairs/s), or where
    // the VALU full-search kernel does not exist (B = 16).  FIC_RGB_SWEEP=1|2 / option "sweep" override.
    int want = c->opt_sweep;
    if (const char* env = getenv("FIC_RGB_SWEEP"))
        if (env[0] >= '1' && env[0] <= '2' && !env[1]) want = env[0] - '0';
    const bool use_q = g.full && g.Nd < (1 << 24) && (want == 2 || (want == 0 && ((double)g.Nr * g.Nd >= 3e7 || g.B == 16)));
    if (use_q && rgb_q_setup(c)) return FIC_E_HIP;
    if (use_q) rgb_q_chunks(c);
    if (!use_q && g.full && g.B <= 8 && !c->pool_cf) {      // the VALU full-search sweep's f32 pool copy (k_sweep_rgb_fast), on first use
        const int rc = dev_alloc(&c->pool_cf, (size_t)g.planes * g.Nd * g.n);
        if (rc != FIC_OK) return rc;
    }
    c->last_sweep = use_q ? 2 : 1;
    {
        FicRgbBuffers b;                                  // image 0 of the batch: the kernels take the image from their grid
        FicRgbOutputs o;
        rgb_plane(c, 0, &b, &o);
        if (fic_launch_rgb_encode(b, o, with_collage ? c->collage : nullptr, g, s, use_q ? &c->q : nullptr))
            return fail(FIC_E_HIP, "RGB kernel launch failed");
    }
    c->encoded_any = true;
    c->have_collage = with_collage != 0;
    return FIC_OK;
}

int fic_rgb_ctx_set_option(fic_rgb_ctx* c, const char* name, int value)
{
    if (!c || !name) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_option: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!strcmp(name, "sweep")) {
        if (value < 0 || value > 2) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_option: sweep must be 0 (auto), 1 (VALU) or 2 (matrix cores)");
        c->opt_sweep = value;
        return FIC_OK;
    }
    if (!strcmp(name, "chunks")) {
        if (value < 0) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_option: chunks must be >= 0");
        c->opt_chunks = value;
        return FIC_OK;
    }
    return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_set_option: unknown option '%s'", name);
}

int fic_rgb_ctx_last_sweep(fic_rgb_ctx* c)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_last_sweep: null context");
    return c->last_sweep;
}

int fic_rgb_ctx_sync(fic_rgb_ctx* c)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    return FIC_OK;
}

int fic_rgb_ctx_get_results_host(fic_rgb_ctx* c, int32_t* idx_local, float* a, float* bR, float* bG, float* bB, int32_t* qrows5,
                                 int32_t* collage_argb)
{
    if (!c) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_get_results_host: null context");
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_rgb_ctx_get_results_host: nothing encoded yet");
    if (collage_argb && !c->have_collage) return fail(FIC_E_STATE, "fic_rgb_ctx_get_results_host: the last encode built no collage");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->last_stream));
    const size_t n = (size_t)c->g.planes * c->g.Nr;
    if (idx_local) HIP_TRY(hipMemcpy(idx_local, c->idx_local, n * 4, hipMemcpyDeviceToHost));
    if (a) HIP_TRY(hipMemcpy(a, c->a, n * 4, hipMemcpyDeviceToHost));
    if (bR) HIP_TRY(hipMemcpy(bR, c->bR, n * 4, hipMemcpyDeviceToHost));
    if (bG) HIP_TRY(hipMemcpy(bG, c->bG, n * 4, hipMemcpyDeviceToHost));
    if (bB) HIP_TRY(hipMemcpy(bB, c->bB, n * 4, hipMemcpyDeviceToHost));
    if (qrows5) HIP_TRY(hipMemcpy(qrows5, c->qrows, n * 20, hipMemcpyDeviceToHost));
    if (collage_argb) HIP_TRY(hipMemcpy(collage_argb, c->collage, (size_t)c->g.planes * c->g.W * c->g.H * 4, hipMemcpyDeviceToHost));
    return FIC_OK;
}

// decodeRGB (FC:430-508) from the context's quantised rows, plane by plane, everything device resident
int fic_rgb_ctx_decode_host(fic_rgb_ctx* c, int32_t* argb_out, float* avg_error_out, int* iterations_out)
{
    if (!c || !argb_out) return fail(FIC_E_ARGUMENT, "fic_rgb_ctx_decode_host: null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->encoded_any) return fail(FIC_E_STATE, "fic_rgb_ctx_decode_host: nothing encoded yet");
    HIP_TRY(hipSetDevice(c->device));
    const FicGeom& g = c->g;
    FicGeom g1 = g;
    g1.planes = 1;
    const size_t npix = (size_t)g.W * g.H;
    if (!c->dec_image) { int rc = dev_alloc(&c->dec_image, npix); if (rc) return rc; }
    if (!c->dec_scaled) { int rc = dev_alloc(&c->dec_scaled, (size_t)g.Ws * g.Hs); if (rc) return rc; }
    if (!c->dec_state) { int rc = dev_alloc(&c->dec_state, 1); if (rc) return rc; }
    if (!c->dec_sq) { int rc = dev_alloc(&c->dec_sq, fic_decode_sq_words(1, npix)); if (rc) return rc; }
    hipStream_t s = c->last_stream;
    std::vector<int32_t> init(npix, (int32_t)0xff808080u);                          // generateGrayImage FC:1142-1148
    for (int p = 0; p < g.planes; p++) {
        FicDecodeState st;
        memset(&st, 0, sizeof(st));
        HIP_TRY(hipMemcpyAsync(c->dec_image, init.data(), npix * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->dec_state, &st, sizeof(st), hipMemcpyHostToDevice, s));
        for (int counter = 0; counter < 50; counter++) {
            if (fic_launch_decode_iteration_rgb(c->dec_scaled, c->dec_image, c->qrows + (size_t)p * g.Nr * 5, c->dec_state, c->dec_sq,
                                                counter, g1, s))
                return fail(FIC_E_HIP, "decodeRGB iteration launch failed");
            if ((counter & 7) == 7 || counter == 49) {
                HIP_TRY(hipMemcpyAsync(&st, c->dec_state, sizeof(st), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                if (st.done) break;
            }
        }
        if (st.bad_index) return fail(FIC_E_ARGUMENT, "decodeRGB: a codebook row of plane %d points outside the domain pool", p);
        HIP_TRY(hipMemcpy(argb_out + (size_t)p * npix, c->dec_image, npix * 4, hipMemcpyDeviceToHost));
        if (avg_error_out) avg_error_out[p] = st.avg_out;
        if (iterations_out) iterations_out[p] = st.iters;
    }
    return FIC_OK;
}

namespace {
fic_rgb_ctx* rgb_cache_take(int device, int w, int h, int B, int wK)
{
    std::lock_guard<std::mutex> lk(g_rgb_mu);
    for (size_t i = g_rgb_cache.size(); i-- > 0;) {
        fic_rgb_ctx* c = g_rgb_cache[i];
        const FicGeom& g = c->g;
        if (c->device == device && g.W == w && g.H == h && g.B == B && g.wK == wK && g.planes == 1) {
            g_rgb_cache.erase(g_rgb_cache.begin() + (long)i);
            return c;
        }
    }
    return nullptr;
}
void rgb_cache_give(fic_rgb_ctx* c)
{
    fic_rgb_ctx* evict = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_rgb_mu);
        g_rgb_cache.push_back(c);
        if (g_rgb_cache.size() > 4) {
            evict = g_rgb_cache.front();
            g_rgb_cache.erase(g_rgb_cache.begin());
        }
    }
    if (evict) fic_rgb_ctx_destroy(evict);
}
}  // namespace

int fic_encode_rgb_argb(const int32_t* argb, int w, int h, int B, int wK, int device, int32_t* idx_local, float* a,
                        float* bR, float* bG, float* bB, int32_t* qrows5, int32_t* collage_argb)
{
    if (!argb || !idx_local || !a || !bR || !bG || !bB) return fail(FIC_E_ARGUMENT, "fic_encode_rgb_argb: null argument");
    // the GUI re-encodes the same image on every slider move (CTL:125-145): keep the last few working sets
    fic_rgb_ctx* c = rgb_cache_take(device, w, h, B, wK);
    if (!c) c = fic_rgb_ctx_create(device, w, h, B, wK, 1);
    if (!c) return g_err_code ? g_err_code : FIC_E_HIP;   // fic_rgb_ctx_create recorded why
    int rc = fic_rgb_ctx_set_argb_host(c, argb);
    if (rc == FIC_OK) rc = fic_rgb_ctx_encode(c, collage_argb ? 1 : 0, nullptr);
    if (rc == FIC_OK) rc = fic_rgb_ctx_get_results_host(c, idx_local, a, bR, bG, bB, qrows5, collage_argb);
    ErrKeep keep;
    if (rc == FIC_OK) rgb_cache_give(c);
    else fic_rgb_ctx_destroy(c);
    return rc;
}

int64_t fic_write_run_rgb(const int32_t* qrows5, int n_ranges, int w, int h, int B, int wK, uint8_t* out, int64_t capacity)
{
    if (!qrows5 || !out || n_ranges < 0) return fail(FIC_E_ARGUMENT, "fic_write_run_rgb: bad argument");
    int64_t need = 20 + 20 * (int64_t)n_ranges;
    if (capacity < need) return fail(FIC_E_CAPACITY, "fic_write_run_rgb: need %lld bytes, have %lld", (long long)need, (long long)capacity);
    auto put = [](uint8_t* p, int32_t v) {
        uint32_t u = (uint32_t)v;
        p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u;
    };
    const int32_t hdr[5] = {1, w, h, B, wK};          // FC:234-238, isRGB = 1
    for (int i = 0; i < 5; i++) put(out + 4 * i, hdr[i]);
    uint8_t* p = out + 20;
    for (int64_t i = 0; i < 5 * (int64_t)n_ranges; i++, p += 4) put(p, qrows5[i]);   // FC:249-256
    return need;
}

}  // extern "C"

void ficd::release_rgb_cache()
{
    std::vector<fic_rgb_ctx*> drop;
    {
        std::lock_guard<std::mutex> lk(g_rgb_mu);
        drop.swap(g_rgb_cache);
    }
    for (fic_rgb_ctx* c : drop) fic_rgb_ctx_destroy(c);
}
