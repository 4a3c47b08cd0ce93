// fic_capi_multi.cpp -- C ABI, the in-library multi-device encode: one synchronous call, one host thread, n devices, RCCL
// gather of the codebook records (SURVEY.md 8b "n_gpus", 8e).  Host-side orchestration only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "fic_internal.h"

using namespace ficd;

extern "C" {

// ---- in-library multi-device encode (SURVEY.md 8b "n_gpus", 8e) -------------------------------------------------
// FractalCompression.encode (FC:54-59) is ONE synchronous call on one host thread; the range loop it replaces
// (FC:125-159) carries no state between iterations, so the call shards its range blocks over n_gpus devices from that
// one thread: one context and one non-blocking stream per device, tile-aligned spans (the rule of sharding.shard_spans),
// every device builds its own replica of the pool from the replicated image, and the 24-byte codebook records are
// gathered to device 0 with ONE grouped RCCL send/recv (latency-bound on xGMI; no ring, no all-reduce).
// RCCL is loaded on first use (librccl.so.1): a single-GPU host needs no RCCL.
namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::vector<int> devs;               // devices of the live communicators
    std::vector<ncclComm_t> comms;
};
std::mutex g_multi_mu;
Rccl g_rccl;
int g_gather_fallbacks = 0;              // calls that finished with peer copies after an RCCL failure

int rccl_load()
{
    if (g_rccl.lib) return FIC_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(FIC_E_HIP, "multi-device encode needs RCCL: %s", dlerror());
#define FIC_RCCL_SYM(field, name)                                                                   \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                                          \
    if (!g_rccl.field) { dlclose(h); return fail(FIC_E_HIP, "librccl lacks %s", name); }
    FIC_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    FIC_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    FIC_RCCL_SYM(GroupStart, "ncclGroupStart")
    FIC_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    FIC_RCCL_SYM(Send, "ncclSend")
    FIC_RCCL_SYM(Recv, "ncclRecv")
    FIC_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef FIC_RCCL_SYM
    g_rccl.lib = h;
    return FIC_OK;
}
void rccl_drop_comms()
{
    for (ncclComm_t c : g_rccl.comms)
        if (c) (void)g_rccl.CommDestroy(c);
    g_rccl.comms.clear();
    g_rccl.devs.clear();
}
// one communicator per device, created once per device list and kept (ncclCommInitAll costs ~a second)
int rccl_comms(const std::vector<int>& devs)
{
    int rc = rccl_load();
    if (rc) return rc;
    if (g_rccl.devs == devs && !g_rccl.comms.empty()) return FIC_OK;
    rccl_drop_comms();
    g_rccl.comms.assign(devs.size(), nullptr);
    ncclResult_t r = g_rccl.CommInitAll(g_rccl.comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
        g_rccl.comms.clear();
        return fail(FIC_E_HIP, "ncclCommInitAll(%d devices): %s", (int)devs.size(), g_rccl.GetErrorString(r));
    }
    g_rccl.devs = devs;
    return FIC_OK;
}

// One piece of the gather: `cnt` int32 from `src` on the device of communicator rank `from` to `dst` on rank 0's device.
struct GatherPiece {
    int from;
    const int32_t* src;
    int32_t* dst;
    size_t cnt;
    hipStream_t s_from, s_to;
};
// ONE grouped send/recv of all pieces over the live communicators.  The group is ALWAYS closed: an error inside
// ncclGroupStart .. ncclGroupEnd that returned early would leave this thread's group open, the next call's GroupStart /
// GroupEnd pair would nest inside it and launch nothing while reporting success.  So the first error is
// remembered, the loop stops issuing, GroupEnd runs regardless, and on any failure the communicators are dropped: the next
// call starts from fresh communicators and group depth 0.  `inject_fail` (test hook): piece 0's send is given a peer that
// does not exist, which RCCL refuses with ncclInvalidArgument.
int rccl_grouped_gather(const std::vector<GatherPiece>& pieces, bool inject_fail)
{
    ncclResult_t first = ncclSuccess;
    hipError_t hfirst = hipSuccess;
    const char* where = "";
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) {
        rccl_drop_comms();
        return fail(FIC_E_HIP, "ncclGroupStart: %s", g_rccl.GetErrorString(r));
    }
    for (size_t i = 0; i < pieces.size() && first == ncclSuccess && hfirst == hipSuccess; i++) {
        const GatherPiece& p = pieces[i];
        if ((hfirst = hipSetDevice(g_rccl.devs[p.from])) != hipSuccess) { where = "hipSetDevice(sender)"; break; }
        const int peer = (inject_fail && i == 0) ? (int)g_rccl.comms.size() + 7 : 0;
        if ((first = g_rccl.Send(p.src, p.cnt, ncclInt32, peer, g_rccl.comms[p.from], p.s_from)) != ncclSuccess) { where = "ncclSend"; break; }
        if ((hfirst = hipSetDevice(g_rccl.devs[0])) != hipSuccess) { where = "hipSetDevice(receiver)"; break; }
        if ((first = g_rccl.Recv(p.dst, p.cnt, ncclInt32, p.from, g_rccl.comms[0], p.s_to)) != ncclSuccess) { where = "ncclRecv"; break; }
    }
    r = g_rccl.GroupEnd();                                   // always: closes the group whatever happened inside it
    if (first == ncclSuccess && hfirst == hipSuccess && r == ncclSuccess) return FIC_OK;
    char msg[256];
    if (hfirst != hipSuccess) snprintf(msg, sizeof(msg), "%s: %s", where, hipGetErrorString(hfirst));
    else if (first != ncclSuccess) snprintf(msg, sizeof(msg), "%s: %s", where, g_rccl.GetErrorString(first));
    else snprintf(msg, sizeof(msg), "ncclGroupEnd: %s", g_rccl.GetErrorString(r));
    rccl_drop_comms();
    return fail(FIC_E_HIP, "codebook gather over RCCL failed (%s); communicators dropped", msg);
}

// gather of the span records [begin_i, begin_i + count_i) of every context's `records` array into context 0's
int gather_rccl(const std::vector<fic_ctx*>& ctx, const std::vector<int>& begin, const std::vector<int>& count)
{
    std::vector<int> devs;
    for (fic_ctx* c : ctx) devs.push_back(c->device);
    int rc = rccl_comms(devs);
    if (rc) return rc;
    std::vector<GatherPiece> pieces;
    for (size_t i = 1; i < ctx.size(); i++) {
        if (count[i] == 0) continue;
        const size_t off = (size_t)begin[i] * 6, cnt = (size_t)count[i] * 6;
        pieces.push_back({(int)i, ctx[i]->o.records + off, ctx[0]->o.records + off, cnt, ctx[i]->own_stream, ctx[0]->own_stream});
    }
    if (pieces.empty()) return FIC_OK;
    return rccl_grouped_gather(pieces, false);
}
// pinned host staging of the grey image for the per-device uploads (grown on demand, freed by fic_release_cache)
uint8_t* g_stage = nullptr;
size_t g_stage_cap = 0;
int stage_reserve(size_t bytes)
{
    if (bytes <= g_stage_cap) return FIC_OK;
    if (g_stage) { (void)hipHostFree(g_stage); g_stage = nullptr; g_stage_cap = 0; }
    HIP_TRY(hipHostMalloc((void**)&g_stage, bytes, hipHostMallocPortable));
    g_stage_cap = bytes;
    return FIC_OK;
}

// the same gather as plain device copies: logical shards that share a physical device (FIC_FAKE_DEVICES, a test knob --
// RCCL refuses two ranks on one device) and FIC_GATHER=copy
int gather_copy(const std::vector<fic_ctx*>& ctx, const std::vector<int>& begin, const std::vector<int>& count)
{
    for (size_t i = 1; i < ctx.size(); i++) {
        if (count[i] == 0) continue;
        const size_t off = (size_t)begin[i] * 6, bytes = (size_t)count[i] * 6 * sizeof(int32_t);
        HIP_TRY(hipSetDevice(ctx[i]->device));
        if (ctx[i]->device == ctx[0]->device)
            HIP_TRY(hipMemcpyAsync(ctx[0]->o.records + off, ctx[i]->o.records + off, bytes, hipMemcpyDeviceToDevice, ctx[i]->own_stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(ctx[0]->o.records + off, ctx[0]->device, ctx[i]->o.records + off, ctx[i]->device, bytes,
                                       ctx[i]->own_stream));
    }
    for (size_t i = 1; i < ctx.size(); i++) {
        HIP_TRY(hipSetDevice(ctx[i]->device));
        HIP_TRY(hipStreamSynchronize(ctx[i]->own_stream));
    }
    return FIC_OK;
}

int encode_multi(const uint8_t* gray, const int32_t* argb, int w, int h, int B, int wK, int n_iso, int n_gpus,
                 int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if ((!gray && !argb) || !idx_local || !a || !b) return fail(FIC_E_ARGUMENT, "fic_encode_gray_multi: null argument");
    if (n_gpus < 1) return fail(FIC_E_ARGUMENT, "n_gpus=%d", n_gpus);
    FicGeom g;
    int rc = make_geometry(w, h, B, wK, n_iso, 1, &g);
    if (rc) return rc;
    const int ndev = fic_device_count();
    if (ndev <= 0) return fail(FIC_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    const char* fake = getenv("FIC_FAKE_DEVICES");        // test knob: logical devices beyond the real ones share them round-robin
    if (n_gpus > ndev && !(fake && atoi(fake) >= n_gpus))
        return fail(FIC_E_NO_DEVICE, "n_gpus=%d but %d HIP device(s) visible", n_gpus, ndev);
    const bool distinct = n_gpus <= ndev;
    const char* gmode = getenv("FIC_GATHER");
    const bool use_rccl = distinct && !(gmode && !strcmp(gmode, "copy"));
    std::lock_guard<std::mutex> lk(g_multi_mu);
    // tile-aligned spans: no sweep tile is computed twice (same rule as sharding.shard_spans)
    const int tsz = 64 * g.NR;
    std::vector<int> begin(n_gpus), count(n_gpus);
    for (int r = 0; r < n_gpus; r++) {
        const long long t0 = (long long)g.tiles * r / n_gpus, t1 = (long long)g.tiles * (r + 1) / n_gpus;
        const int bb = (int)(t0 * tsz < g.Nr ? t0 * tsz : g.Nr), ee = (int)(t1 * tsz < g.Nr ? t1 * tsz : g.Nr);
        begin[r] = bb;
        count[r] = ee - bb;
    }
    std::vector<fic_ctx*> ctx(n_gpus, nullptr);
    auto drop = [&](bool keep) {
        ErrKeep keep_err;
        for (fic_ctx* c : ctx)
            if (c) { if (keep) cache_give(c); else fic_ctx_destroy(c); }
    };
    // The image goes to every device (each builds its own pool replica: cheaper than shipping the 16x-expanded pool).  ONE
    // pinned staging copy of the grey bytes (the R channel of ARGB input, FC:596/977), then one hipMemcpyAsync per device on
    // that device's own stream, the encode queued right behind it: no device waits for another device's upload.
    const size_t npix = (size_t)w * h;
    rc = stage_reserve(npix);
    if (rc == FIC_OK) {
        if (gray) memcpy(g_stage, gray, npix);
        else for (size_t i = 0; i < npix; i++) g_stage[i] = (uint8_t)((argb[i] >> 16) & 0xff);
    }
    for (int i = 0; i < n_gpus && rc == FIC_OK; i++) {
        const int dev = i % ndev;
        ctx[i] = cache_take(dev, w, h, B, wK, n_iso);
        if (!ctx[i]) ctx[i] = fic_ctx_create(dev, w, h, B, wK, n_iso, 1);
        if (!ctx[i]) { rc = g_err_code ? g_err_code : FIC_E_HIP; break; }
        fic_ctx* c = ctx[i];
        hipError_t e = hipSetDevice(dev);
        if (e == hipSuccess && !c->own_stream) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
        if (e == hipSuccess && !c->gray_own) e = hipMalloc((void**)&c->gray_own, npix);
        if (e == hipSuccess && c->last_stream != c->own_stream) e = hipStreamSynchronize(c->last_stream);   // a parked context last used elsewhere
        if (e == hipSuccess) e = hipMemcpyAsync(c->gray_own, g_stage, npix, hipMemcpyHostToDevice, c->own_stream);
        if (e != hipSuccess) { rc = fail(FIC_E_HIP, "image upload to device %d: %s", dev, hipGetErrorString(e)); break; }
        c->b.gray = c->gray_own;
        c->have_input = true;
        c->last_stream = c->own_stream;
    }
    for (int i = 0; i < n_gpus && rc == FIC_OK; i++)
        rc = fic_ctx_encode(ctx[i], begin[i], count[i], ctx[i]->own_stream);      // a rank without tiles: count 0, nothing launched
    if (rc == FIC_OK) {
        // FIC_GATHER: unset / "rccl" = RCCL, and if RCCL fails (the communicators are dropped, see rccl_grouped_gather) this
        // call finishes with peer copies; "rccl-only" = no fallback (the error is returned); "copy" = peer copies only
        const bool strict = gmode && !strcmp(gmode, "rccl-only");
        if (use_rccl) {
            rc = gather_rccl(ctx, begin, count);
            if (rc != FIC_OK && !strict) {
                g_gather_fallbacks++;
                rc = gather_copy(ctx, begin, count);
            }
        } else {
            rc = gather_copy(ctx, begin, count);
        }
    }
    std::vector<int32_t> rec;
    if (rc == FIC_OK) {
        rec.resize((size_t)g.Nr * 6);
        hipError_t e = hipSetDevice(ctx[0]->device);
        if (e == hipSuccess) e = hipMemcpyAsync(rec.data(), ctx[0]->o.records, rec.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx[0]->own_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx[0]->own_stream);
        if (e != hipSuccess) rc = fail(FIC_E_HIP, "codebook copy: %s", hipGetErrorString(e));
    }
    if (rc != FIC_OK) {
        // leave no stream with work in flight behind a failed call
        for (fic_ctx* c : ctx)
            if (c && c->own_stream) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->own_stream); }
        drop(false);
        return rc;
    }
    for (int j = 0; j < g.Nr; j++) {
        const int32_t* r6 = &rec[(size_t)j * 6];
        idx_local[j] = r6[0];
        memcpy(&a[j], &r6[1], 4);
        memcpy(&b[j], &r6[2], 4);
        if (iso) iso[j] = r6[3];
        if (qrows) { qrows[3 * j] = r6[0]; qrows[3 * j + 1] = r6[4]; qrows[3 * j + 2] = r6[5]; }
    }
    for (fic_ctx* c : ctx)          // ranks without tiles may still have their upload in flight: nothing reads the staging buffer after return
        if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->own_stream); }
    drop(true);
    return FIC_OK;
}

}  // namespace

// Test hook: loads RCCL, creates (and keeps) one communicator per device 0..|n|-1 and, for |n| >= 2, runs the gather's
// grouped send/recv pattern (rccl_grouped_gather, the function the encode uses) on 6-int records and checks what arrives.
// n < 0: the ERROR path of the gather on |n| devices -- one send is given a peer that does not exist; the call must fail,
// leave no group open and no communicator cached, and the same pattern must succeed right afterwards on fresh
// communicators.  Returns FIC_OK or a negative code.
int fic_debug_rccl_selftest(int n_signed)
{
    const bool inject = n_signed < 0;
    const int n = inject ? -n_signed : n_signed;
    const int ndev = fic_device_count();
    if (n < 1 || n > ndev) return fail(FIC_E_NO_DEVICE, "fic_debug_rccl_selftest: %d of %d devices", n, ndev);
    std::lock_guard<std::mutex> lk(g_multi_mu);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = i;
    int rc = rccl_comms(devs);
    if (rc) return rc;
    std::vector<int32_t*> buf(n, nullptr);
    std::vector<hipStream_t> st(n, nullptr);
    const size_t cnt = 6 * 1000;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; i++) {
        e = hipSetDevice(i);
        if (e == hipSuccess) e = hipMalloc((void**)&buf[i], cnt * n * 4);
        if (e == hipSuccess) e = hipMemset(buf[i], i + 1, cnt * n * 4);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);       // the fill is only enqueued; the sends run on non-blocking streams
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    if (e != hipSuccess) rc = fail(FIC_E_HIP, "rccl selftest setup: %s", hipGetErrorString(e));
    auto pattern = [&]() {
        std::vector<GatherPiece> pieces;
        for (int i = 1; i < n; i++) pieces.push_back({i, buf[i] + cnt * i, buf[0] + cnt * i, cnt, st[i], st[0]});
        return pieces;
    };
    if (rc == FIC_OK && inject) {
        std::vector<GatherPiece> pieces = pattern();
        if (pieces.empty()) pieces.push_back({0, buf[0], buf[0], cnt, st[0], st[0]});     // one device: a send that can only fail
        const int bad = rccl_grouped_gather(pieces, true);
        if (bad == FIC_OK) rc = fail(FIC_E_HIP, "rccl selftest: a send to a peer that does not exist was accepted");
        else if (!g_rccl.comms.empty()) rc = fail(FIC_E_HIP, "rccl selftest: communicators still cached after a failed gather");
        else rc = rccl_comms(devs);                                                        // fresh communicators, group depth 0
        if (rc == FIC_OK && n == 1) {
            // an empty group on the fresh communicator: fails if the failed call had left a group open
            ncclResult_t r = g_rccl.GroupStart();
            if (r == ncclSuccess) r = g_rccl.GroupEnd();
            if (r != ncclSuccess) rc = fail(FIC_E_HIP, "rccl selftest: group after a failed gather: %s", g_rccl.GetErrorString(r));
        }
    }
    if (rc == FIC_OK && n >= 2) {
        rc = rccl_grouped_gather(pattern(), false);
        std::vector<int32_t> host(cnt * n);
        if (rc == FIC_OK) {
            (void)hipSetDevice(0);
            e = hipStreamSynchronize(st[0]);
            if (e == hipSuccess) e = hipMemcpy(host.data(), buf[0], host.size() * 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(FIC_E_HIP, "rccl selftest readback: %s", hipGetErrorString(e));
            for (int i = 0; i < n && rc == FIC_OK; i++) {
                const int32_t want = 0x01010101 * (i + 1);
                for (size_t k = 0; k < cnt; k++)
                    if (host[cnt * i + k] != want) { rc = fail(FIC_E_HIP, "rccl selftest: wrong data from device %d", i); break; }
            }
        }
    }
    {
        ErrKeep keep;
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(i);
            if (st[i]) { (void)hipStreamSynchronize(st[i]); (void)hipStreamDestroy(st[i]); }
            if (buf[i]) (void)hipFree(buf[i]);
        }
    }
    return rc;
}

// Test hook / diagnostics: calls of the multi-device entry that finished with peer copies after an RCCL failure.
int fic_debug_gather_fallbacks(void) { return g_gather_fallbacks; }

int fic_encode_gray_argb_multi(const int32_t* argb, int w, int h, int B, int wK, int n_iso, int n_gpus, int32_t* idx_local,
                               float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if (n_gpus == 1) return encode_oneshot(nullptr, argb, w, h, B, wK, n_iso, 0, idx_local, a, b, iso, qrows);
    return encode_multi(nullptr, argb, w, h, B, wK, n_iso, n_gpus, idx_local, a, b, iso, qrows);
}

int fic_encode_gray_u8_multi(const uint8_t* gray, int w, int h, int B, int wK, int n_iso, int n_gpus, int32_t* idx_local,
                             float* a, float* b, int32_t* iso, int32_t* qrows)
{
    if (n_gpus == 1) return encode_oneshot(gray, nullptr, w, h, B, wK, n_iso, 0, idx_local, a, b, iso, qrows);
    return encode_multi(gray, nullptr, w, h, B, wK, n_iso, n_gpus, idx_local, a, b, iso, qrows);
}

}  // extern "C"

void ficd::release_comms()
{
    std::lock_guard<std::mutex> lk(g_multi_mu);
    if (g_rccl.lib) rccl_drop_comms();
    if (g_stage) { (void)hipHostFree(g_stage); g_stage = nullptr; g_stage_cap = 0; }
}
