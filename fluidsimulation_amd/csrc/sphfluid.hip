// sphfluid.hip -- libsphfluid.so: host side of the C ABI in include/sphfluid.h.
//
// Owns the device memory, the stream and the per-frame sequence; the kernels live in
// sph_sort.h / sph_neighbors.h / sph_stream.h.  gfx950 only; there is no CPU path.
#include "../../include/sphfluid.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: the library is opened with dlopen in sph_slab_comm_init
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "sph_device.h"
#include "sph_neighbors.h"
#include "sph_sort.h"
#include "sph_stream.h"
#include "sph_slab.h"

static_assert(sizeof(SphParticle) == 16, "FluidParticleComponent layout");
static_assert(sizeof(SphBoundaryParticle) == 12, "BoundaryParticleComponent layout");
static_assert(sizeof(SphBox) == 25 * 4, "SphBox layout");

namespace {

thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a, b;
    int stage;
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
};

struct MatKey {
    uint32_t w[5];
    bool operator==(const MatKey &o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
struct MatKeyHash {
    size_t operator()(const MatKey &k) const
    {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 5; i++) h = (h ^ k.w[i]) * 1099511628211ull;
        return (size_t)h;
    }
};

} // namespace

#define SPH_QUEUE_SLOTS 64 // tileCtl: NB_CTL_* words and work-queue counters, cleared every step;
                           // one more word behind them (SPH_CTL_ERROR_WORD) is the sticky error flag

// ---- scratch of sph_sort_pairs_noalloc ------------------------------------------------------
// The one entry point that allocates nothing keeps its scratch (one ping-pong (key, value) buffer,
// the per-tile digit histograms and the digit totals of the radix sort) in this file-scope device
// array.  Every GPU the library runs on holds its own copy until the process ends; `scratch` is
// only a host-side stand-in for it, the device address is taken once per device in sph_create
// (hipGetSymbolAddress with that device current) and kept in g_scratchPtr.
struct SortScratch {
    uint32_t keys[SPH_SORT_NOALLOC_MAX];
    uint32_t vals[SPH_SORT_NOALLOC_MAX];
    uint32_t hist[256 * ((SPH_SORT_NOALLOC_MAX + RS_TILE - 1) / RS_TILE)];
    uint32_t rowTotal[256];
};
static __device__ __align__(256) unsigned char scratch[sizeof(SortScratch)];
#define SPH_MAX_DEVICES 64
static SortScratch *g_scratchPtr[SPH_MAX_DEVICES]; // per device; written under g_scratchLock
static std::mutex g_scratchLock;

struct Stage {
    static constexpr size_t CHUNK = 16u << 20;
    static constexpr int SLOTS = 3;
    char *buf[SLOTS] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr};
    bool busy[SLOTS] = {false, false, false}; // an H2D copy out of the slot may still be running
    // worker threads: copy part i of the current job
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cvWork, cvDone;
    char *dst = nullptr;
    const char *src = nullptr;
    size_t len = 0;
    int parts = 0, taken = 0, done = 0;
    uint64_t gen = 0;
    bool quit = false;

    static void copy_part(char *d, const char *s0, size_t n, int part, int parts)
    {
        const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
        const size_t lo = std::min(n, per * (size_t)part), hi = std::min(n, lo + per);
        if (hi > lo) std::memcpy(d + lo, s0 + lo, hi - lo);
    }
    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cvWork.wait(lk, [&] { return quit || (gen != seen && taken < parts); });
            if (quit) return;
            while (taken < parts) {
                const int part = taken++;
                char *d = dst;
                const char *s0 = src;
                const size_t n = len;
                const int np = parts;
                lk.unlock();
                copy_part(d, s0, n, part, np);
                lk.lock();
                if (++done == parts) cvDone.notify_all();
            }
            seen = gen;
        }
    }
    // memcpy by the calling thread and the workers together
    void copy(char *d, const char *s0, size_t n)
    {
        if (n < (2u << 20) || th.empty()) {
            std::memcpy(d, s0, n);
            return;
        }
        std::unique_lock<std::mutex> lk(m);
        dst = d; src = s0; len = n;
        parts = (int)th.size() + 1;
        taken = done = 0;
        gen++;
        cvWork.notify_all();
        while (taken < parts) {
            const int part = taken++;
            lk.unlock();
            copy_part(d, s0, n, part, parts);
            lk.lock();
            ++done;
        }
        cvDone.wait(lk, [&] { return done == parts; });
    }
    ~Stage()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cvWork.notify_all();
        for (auto &t : th) t.join();
        for (int i = 0; i < SLOTS; i++) {
            if (buf[i]) (void)hipHostFree(buf[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
};

struct SphCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    // second stream: the neighbour plan of a step runs beside the reorder pass (sph_step_resident)
    hipStream_t stream2 = nullptr;
    hipEvent_t evTable = nullptr, evPlan = nullptr;
    hipDeviceProp_t prop;
    std::string err;
    int debug = 0;
    int maxCells = 1 << 25;

    // fluid
    int n = 0;
    int cur = 0; // which of the double buffers holds the current state
    DevBuf<float4> posm[2], velq[2];
    DevBuf<uint32_t> mat[2], oid[2];
    DevBuf<uint32_t> keysA, keysB, valsA, valsB;
    DevBuf<float> rho, prs;
    DevBuf<float4> force;
    DevBuf<int> dcount, fcount;
    DevBuf<float4> zeros; // 16 zero records: source of the boundary's (v, q) in the force tile
    DevBuf<uint32_t> hashes, keysEcs;
    DevBuf<int> cells;
    DevBuf<float> io3a, io3b, io1;
    DevBuf<uint32_t> ioU;
    uint32_t *sortedKeys = nullptr; // points into keysA/keysB
    uint32_t *sortedVals = nullptr;
    bool haveKeys = false, haveSort = false, haveTable = false, ordered = false;
    bool haveDensity = false, haveForce = false;
    bool rhoInOrder = false; // rho[] lines up with the current particle arrays

    // materials
    std::vector<MatKey> matKeys;
    std::vector<Mat> mats;
    float matRate = 0.f;
    DevBuf<Mat> dmats;
    std::unordered_map<MatKey, uint32_t, MatKeyHash> matIndex;
    std::vector<uint32_t> hostMatIdx;
    bool matUniform = false; // every particle of the last upload carries material 0
    bool matAllZero = false; // ... and both material-index buffers hold zeros: reorder and integrate leave them alone
    // SphConfig.pinHostArrays: caller arrays registered with the driver (page-locked) on first
    // use and kept, so that the per-frame copies of the host round trips run as DMA at PCIe rate
    bool pinHost = false;
    struct Pinned { const void *p; size_t bytes /* registered, 0 = registration failed */, asked; };
    std::vector<Pinned> pinned;
    int pinOk = 0, pinFailed = 0;
    // Staged host copies (arrays that are not registered): a ring of pinned chunks owned by the
    // library, filled / emptied by a few host threads while the DMA of the chunk before runs
    Stage *stage = nullptr;
    bool stageFailed = false;
    std::string pinErr; // text of the last failed registration

    // grid
    GridDesc grid{};
    bool gridExplicit = false, gridValid = false;
    DevBuf<int2> table, btable, ttable;
    DevBuf<uint32_t> tileList, tileCtl; // non-empty tiles; NB_CTL_* words and work-queue counters
    int nbGrid[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // persistent grid of k_neighbors<FORCE,COUNT,MULTI>
    int queueNext = NB_CTL_QUEUE0;       // next unused pair of work-queue words of tileCtl
    // neighbour plan (k_plan): block list, plans, crowded cells
    DevBuf<BlockHdr> blockList;
    DevBuf<int2> gcellList;
    DevBuf<uint32_t> gmaskOff; // one word per entry of gcellList
    int planGrid = 1;          // workgroups of the last k_plan launch
    DevBuf<uint32_t> cellCounts, cellSums; // counting sort of the step entry points (sph_sort.h)
    // slab mode: the integrate kernel counts what the next exchange will send, for the cuts of the
    // last one (fixed cuts: the exchange then skips its counting pass)
    DevBuf<uint32_t> slabNext;
    float expectLo = 0.f, expectHi = 0.f, expectHalo = -1.f;
    bool nextCounts = false;   // slabNext holds the counters for (expectLo, expectHi, expectHalo)
    bool slabCountPass = false; // SPHFLUID_SLAB_COUNT_PASS=1: always count in a pass of its own (tests)
    bool slabKeys = false;     // slab mode: keysA holds this step's keys (written by the pack and import kernels)
    bool radixInSteps = false;  // SPHFLUID_STEP_SORT=radix: the step entry points sort like sph_radix_sort
    DevBuf<uint4> pool;
    bool havePlan = false;
    bool haveMasks = false; // the production density pass has run on this plan
    int fallbackCap = -1; // debug: LDS record cap override for the neighbour kernels
    int cap1 = NB_CAP;    // record cap of single-cell blocks (SPHFLUID_CAP1, experiments)
    long long poolLimit = 0; // debug (SPHFLUID_POOL_LIMIT, 16-byte units): cap on the plan pool, for the overflow test

    // boundary
    int nb = 0;
    DevBuf<float> bpos3, bpar3;
    DevBuf<float4> bposmIn, bposm;
    DevBuf<uint32_t> bkeysA, bkeysB, bvalsA, bvalsB;
    uint32_t *bperm = nullptr;
    bool boundaryDirty = true;
    float hostBMin[3] = {0, 0, 0}, hostBMax[3] = {0, 0, 0};

    // colliders
    int nboxes = 0;
    bool anyDynamicBox = false;
    DevBuf<float> boxState;
    DevBuf<DevBox> devBoxes;
    DevBuf<float4> boxAabb, chunkAabb; // world bounds per box / per SPH_BOX_CHUNK boxes
    DevBuf<float> bodyVelA, bodyVelB, bodyNc; // box-box contact passes: velocities (6 a body), contact counts
    DevBuf<float> bodyImpulse;

    // sort scratch
    DevBuf<uint32_t> hist, rowTotal;
    DevBuf<float> partial;

    // slab decomposition (sph_slab_*)
    bool slab = false;
    int slabCap = 0;    // particle capacity of the fluid buffers
    int nOwned = 0;     // particles this rank owns
    int nGhost = 0;     // live ghost records in [0, n)
    int nDead = 0;      // dead records in [0, n), cut off by the next sort
    DevBuf<uint32_t> slabCounters;
    // native exchange (sph_slab_comm_init / sph_slab_exchange): RCCL communicator, send / receive
    // buffers of 32-byte records for the two x neighbours, headers (4 words a side)
    ncclComm_t comm = nullptr;
    int commRank = 0, commWorld = 0;
    DevBuf<float4> xsend[2], xrecv[2];
    DevBuf<uint32_t> xhdr;          // [0..3] recv from left, [4..7] recv from right
    uint32_t *xhost = nullptr;      // pinned: own counters (4), headers received (8)

    // timings
    int timing = 0; // SphConfig::enableTimings: 0 off, 1 every stage, 2 the neighbour kernels only
    std::vector<hipEvent_t> freeEvents;
    std::vector<EventPair> pending;
    SphTimings tm{};
};

namespace {

int fail(SphCtx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, e_ == hipErrorOutOfMemory ? SPH_E_NOMEM : SPH_E_DEVICE, "%s: %s",     \
                        #expr, hipGetErrorString(e_));                                           \
    } while (0)

template <typename T> int ensure(SphCtx *c, DevBuf<T> &b, size_t count, bool grow = true)
{
    if (count <= b.cap && b.p) return SPH_OK;
    size_t want = grow ? std::max(count, b.cap + b.cap / 2) : count;
    if (want == 0) want = 1;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    HIPCHK(c, hipMalloc((void **)&b.p, want * sizeof(T)));
    b.cap = want;
    return SPH_OK;
}
template <typename T> void release(DevBuf<T> &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

inline int blocks_for(long n, int per) { return (int)((n + per - 1) / per); }

struct StageTimer {
    SphCtx *c;
    EventPair ep;
    bool on;
    hipStream_t s;
    StageTimer(SphCtx *ctx, int stage, hipStream_t stream = nullptr)
        : c(ctx), on(ctx->timing == 1 || (ctx->timing == 2 && (stage == SPH_T_DENSITY || stage == SPH_T_FORCE))),
          s(stream ? stream : ctx->stream)
    {
        if (!on) return;
        for (hipEvent_t *e : {&ep.a, &ep.b}) {
            if (!c->freeEvents.empty()) {
                *e = c->freeEvents.back();
                c->freeEvents.pop_back();
            } else if (hipEventCreate(e) != hipSuccess) {
                on = false;
                return;
            }
        }
        ep.stage = stage;
        (void)hipEventRecord(ep.a, s);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(ep.b, s);
        c->pending.push_back(ep);
    }
};

// A host that enables timings and never reads them must not grow the event pool without bound:
// once a few hundred pairs are pending, the finished ones (in stream order) are folded into the
// totals and their events recycled.
void drain_timings(SphCtx *c)
{
    if (c->pending.size() < 512) return;
    size_t done = 0;
    while (done < c->pending.size() && hipEventQuery(c->pending[done].b) == hipSuccess) {
        const EventPair &ep = c->pending[done];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            c->tm.ms[ep.stage] += ms;
            c->tm.launches[ep.stage] += 1;
        }
        c->freeEvents.push_back(ep.a);
        c->freeEvents.push_back(ep.b);
        done++;
    }
    c->pending.erase(c->pending.begin(), c->pending.begin() + (long)done);
}

int collect_timings(SphCtx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    for (const EventPair &ep : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            c->tm.ms[ep.stage] += ms;
            c->tm.launches[ep.stage] += 1;
        }
        c->freeEvents.push_back(ep.a);
        c->freeEvents.push_back(ep.b);
    }
    c->pending.clear();
    return SPH_OK;
}

// ---- staged host <-> device copies ----------------------------------------------------------
// The span FluidSPHByGPUSystem.cs:227-233,249-252 (SetData / readback of the packed arrays) is a
// copy of caller-owned, pageable memory.  Registering the caller's arrays (SphConfig.pinHostArrays)
// is the zero-copy way, but a driver may refuse it (this pool's boxes do: hipHostRegister returns
// "invalid argument" for any pointer), and the runtime's own pageable path is a single-threaded
// copy through one bounce buffer (17-24 GB/s up, 10 GB/s down here).  So large copies of
// unregistered arrays go through a ring of pinned chunks of the library's own: a few host threads
// fill (or empty) one chunk while the DMA engine moves the one before.
// the ring, made on first use; nullptr when it cannot be had (the copies then go the plain way)
Stage *stage_of(SphCtx *c)
{
    if (c->stage) return c->stage;
    if (c->stageFailed) return nullptr;
    Stage *st = new (std::nothrow) Stage();
    bool ok = st != nullptr;
    for (int i = 0; ok && i < Stage::SLOTS; i++)
        ok = hipHostMalloc((void **)&st->buf[i], Stage::CHUNK, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&st->ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        delete st;
        c->stageFailed = true;
        return nullptr;
    }
    int nt = 4;
    if (const char *e = getenv("SPHFLUID_STAGE_THREADS")) nt = atoi(e);
    nt = std::max(1, std::min(nt, (int)std::max(1u, std::thread::hardware_concurrency())));
    for (int i = 1; i < nt; i++) st->th.emplace_back([st] { st->worker(); });
    c->stage = st;
    return st;
}

bool is_registered(const SphCtx *c, const void *p)
{
    for (const auto &e : c->pinned)
        if (e.p == p && e.bytes) return true;
    return false;
}

// host -> device, asynchronous on the context's stream as far as the device is concerned; the host
// array has been read in full when this returns
int copy_h2d(SphCtx *c, void *dstDev, const void *srcHost, size_t bytes)
{
    Stage *st = (bytes >= (8u << 20) && !is_registered(c, srcHost)) ? stage_of(c) : nullptr;
    if (!st) {
        HIPCHK(c, hipMemcpyAsync(dstDev, srcHost, bytes, hipMemcpyHostToDevice, c->stream));
        return SPH_OK;
    }
    int k = 0;
    for (size_t off = 0; off < bytes; off += Stage::CHUNK, k++) {
        const int slot = k % Stage::SLOTS;
        const size_t n = std::min(Stage::CHUNK, bytes - off);
        if (st->busy[slot]) HIPCHK(c, hipEventSynchronize(st->ev[slot]));
        st->copy(st->buf[slot], (const char *)srcHost + off, n);
        HIPCHK(c, hipMemcpyAsync((char *)dstDev + off, st->buf[slot], n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(st->ev[slot], c->stream));
        st->busy[slot] = true;
    }
    return SPH_OK;
}

// device -> host; the host array is complete when this returns (it synchronises)
int copy_d2h(SphCtx *c, void *dstHost, const void *srcDev, size_t bytes)
{
    Stage *st = (bytes >= (8u << 20) && !is_registered(c, dstHost)) ? stage_of(c) : nullptr;
    if (!st) {
        HIPCHK(c, hipMemcpyAsync(dstHost, srcDev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return SPH_OK;
    }
    const int nchunks = (int)((bytes + Stage::CHUNK - 1) / Stage::CHUNK);
    auto issue = [&](int k) -> hipError_t {
        const int slot = k % Stage::SLOTS;
        const size_t off = (size_t)k * Stage::CHUNK, n = std::min(Stage::CHUNK, bytes - off);
        if (st->busy[slot]) { // (an upload out of the slot)
            const hipError_t e = hipEventSynchronize(st->ev[slot]);
            if (e != hipSuccess) return e;
            st->busy[slot] = false;
        }
        hipError_t e = hipMemcpyAsync(st->buf[slot], (const char *)srcDev + off, n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(st->ev[slot], c->stream);
        return e;
    };
    // two chunks in flight while the third is being emptied
    for (int k = 0; k < std::min(nchunks, Stage::SLOTS - 1); k++) HIPCHK(c, issue(k));
    for (int k = 0; k < nchunks; k++) {
        const int slot = k % Stage::SLOTS;
        const size_t off = (size_t)k * Stage::CHUNK, n = std::min(Stage::CHUNK, bytes - off);
        HIPCHK(c, hipEventSynchronize(st->ev[slot]));
        if (k + Stage::SLOTS - 1 < nchunks) HIPCHK(c, issue(k + Stage::SLOTS - 1));
        st->copy((char *)dstHost + off, st->buf[slot], n);
    }
    return SPH_OK;
}

// ---- materials ----------------------------------------------------------------------------
// constants in float32, in the reference's evaluation order (FluidSPHByCPUSystem.cs:53-60,
// 133-135,224-231); math.PI = 3.14159274f
Mat make_mat(const MatKey &k, float rate)
{
    float radius, restDensity, viscosity, gasConstant, invMass;
    std::memcpy(&radius, &k.w[0], 4);
    std::memcpy(&restDensity, &k.w[1], 4);
    std::memcpy(&viscosity, &k.w[2], 4);
    std::memcpy(&gasConstant, &k.w[3], 4);
    std::memcpy(&invMass, &k.w[4], 4);
    const float PI = 3.14159274f;
    Mat m{};
    m.radius = radius;
    m.restDensity = restDensity;
    m.viscosity = viscosity;
    m.gasConstant = gasConstant;
    m.invMass = invMass;
    m.mass = 1.0f / invMass;
    m.h = radius * rate;
    m.h2 = m.h * m.h;
    volatile float h9 = powf(m.h, 9.0f);
    volatile float den = 64.0f * PI * h9;
    m.poly6 = m.mass * 315.0f / den;
    volatile float h6 = powf(m.h, 6.0f);
    volatile float den6 = PI * h6;
    m.spiky = 45.0f / den6;
    // least t with sqrtf(t) >= h  (sqrtf is correctly rounded and monotone)
    float t = m.h2;
    for (int i = 0; i < 8 && t > 0.f && sqrtf(nextafterf(t, 0.f)) >= m.h; i++) t = nextafterf(t, 0.f);
    for (int i = 0; i < 8 && sqrtf(t) < m.h; i++) t = nextafterf(t, INFINITY);
    m.thrF = t;
    m.negInvH = -1.0f / m.h;
    m.negInvH2 = -1.0f / m.h2;
    m.poly6h6 = m.poly6 * (m.h2 * m.h2 * m.h2);
    return m;
}

int upload_mats(SphCtx *c, float rate)
{
    c->mats.resize(c->matKeys.size());
    for (size_t i = 0; i < c->matKeys.size(); i++) c->mats[i] = make_mat(c->matKeys[i], rate);
    c->matRate = rate;
    int rc = ensure(c, c->dmats, std::max<size_t>(c->mats.size(), 1));
    if (rc) return rc;
    if (!c->mats.empty())
        HIPCHK(c, hipMemcpyAsync(c->dmats.p, c->mats.data(), c->mats.size() * sizeof(Mat),
                                 hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

// deduplicate (params, invMass) tuples into the material table; fills c->hostMatIdx
void build_materials(SphCtx *c, const float *invMass, const SphParticle *params, int n)
{
    c->matKeys.clear();
    c->matIndex.clear();
    c->matUniform = false;
    // the usual frame: one fluid, every particle with the same five numbers.  A scan in a few
    // threads settles that (memory-bound, ~20 B per particle); the index array is then all zeros
    // and is neither built nor uploaded.
    if (n >= (1 << 16)) {
        const int nt = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<int> same((size_t)nt, 1);
        std::vector<std::thread> th;
        const SphParticle p0 = params[0];
        const float m0 = invMass[0];
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                const size_t lo = (size_t)n * t / nt, hi = (size_t)n * (t + 1) / nt;
                bool ok = true;
                for (size_t i = lo; i < hi && ok; i++)
                    ok = std::memcmp(&params[i], &p0, 16) == 0 && std::memcmp(&invMass[i], &m0, 4) == 0;
                same[(size_t)t] = ok ? 1 : 0;
            });
        for (auto &x : th) x.join();
        if (std::all_of(same.begin(), same.end(), [](int v) { return v == 1; })) {
            MatKey k;
            std::memcpy(&k.w[0], &p0, 16);
            std::memcpy(&k.w[4], &m0, 4);
            c->matKeys.push_back(k);
            c->matIndex.emplace(k, 0u);
            c->matUniform = true;
            return;
        }
    }
    c->hostMatIdx.resize((size_t)n);
    MatKey last{};
    uint32_t lastIdx = 0;
    bool haveLast = false;
    for (int i = 0; i < n; i++) {
        MatKey k;
        std::memcpy(&k.w[0], &params[i], 16);
        std::memcpy(&k.w[4], &invMass[i], 4);
        if (haveLast && k == last) {
            c->hostMatIdx[i] = lastIdx;
            continue;
        }
        auto it = c->matIndex.find(k);
        uint32_t idx;
        if (it == c->matIndex.end()) {
            idx = (uint32_t)c->matKeys.size();
            c->matKeys.push_back(k);
            c->matIndex.emplace(k, idx);
        } else {
            idx = it->second;
        }
        c->hostMatIdx[i] = idx;
        last = k;
        lastIdx = idx;
        haveLast = true;
    }
}

// ---- grid ---------------------------------------------------------------------------------
int key_bits(int cells)
{
    int b = 1;
    while ((1ll << b) < cells) b++;
    return b;
}

void set_grid(SphCtx *c, const int origin[3], const int dims[3], float cellSize)
{
    c->slabKeys = false; // (keys of another grid)
    GridDesc g;
    g.ox = origin[0]; g.oy = origin[1]; g.oz = origin[2];
    // dimensions are rounded up to whole tiles
    g.gx = (dims[0] + SPH_TX - 1) / SPH_TX * SPH_TX;
    g.gy = (dims[1] + SPH_TY - 1) / SPH_TY * SPH_TY;
    g.gz = (dims[2] + SPH_TZ - 1) / SPH_TZ * SPH_TZ;
    g.cellSize = cellSize;
    g.cells = g.gx * g.gy * g.gz;
    g.ntx = g.gx / SPH_TX; g.nty = g.gy / SPH_TY; g.ntz = g.gz / SPH_TZ;
    g.tiles = g.ntx * g.nty * g.ntz;
    const GridDesc &o = c->grid;
    if (!c->gridValid || o.ox != g.ox || o.oy != g.oy || o.oz != g.oz || o.gx != g.gx ||
        o.gy != g.gy || o.gz != g.gz || o.cellSize != g.cellSize)
        c->boundaryDirty = true;
    c->grid = g;
    c->gridValid = true;
}

int bounds_of(SphCtx *c, const float4 *p, int n, float mn[3], float mx[3])
{
    const int nb = std::max(1, std::min(1024, blocks_for(n, 256)));
    int rc = ensure(c, c->partial, (size_t)6 * 1024, false);
    if (rc) return rc;
    k_bounds_partial<<<nb, 256, 0, c->stream>>>(p, n, c->partial.p);
    std::vector<float> h((size_t)6 * nb);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->partial.p, h.size() * 4, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int a = 0; a < 3; a++) {
        mn[a] = 3.402823466e+38f;
        mx[a] = -3.402823466e+38f;
    }
    for (int b = 0; b < nb; b++)
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], h[6 * b + a]);
            mx[a] = std::max(mx[a], h[6 * b + 3 + a]);
        }
    return SPH_OK;
}

// size the periodic table from the bounds of fluid + boundary, with a margin of two cells; when
// the box would exceed maxCells the longest axes are cut (cells then alias, which is harmless:
// aliased candidates fail the distance test exactly as hash collisions do in the reference)
int auto_grid(SphCtx *c, float cellSize)
{
    float mn[3], mx[3];
    int rc = bounds_of(c, c->posm[c->cur].p, c->n, mn, mx);
    if (rc) return rc;
    if (c->nb > 0)
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], c->hostBMin[a]);
            mx[a] = std::max(mx[a], c->hostBMax[a]);
        }
    int lo[3], hi[3], dims[3], origin[3];
    for (int a = 0; a < 3; a++) {
        if (!(mn[a] <= mx[a]) || !std::isfinite(mn[a]) || !std::isfinite(mx[a])) {
            mn[a] = 0.f;
            mx[a] = 0.f;
        }
        double l = std::floor((double)mn[a] / cellSize), h = std::floor((double)mx[a] / cellSize);
        l = std::max(l, -1.0e9);
        h = std::min(h, 1.0e9);
        lo[a] = (int)l;
        hi[a] = (int)h;
    }
    if (c->gridValid && !c->gridExplicit && c->grid.cellSize == cellSize) {
        // keep the current table while the particles still fit inside it
        const GridDesc &g = c->grid;
        const int o[3] = {g.ox, g.oy, g.oz}, d[3] = {g.gx, g.gy, g.gz};
        bool fits = true;
        for (int a = 0; a < 3; a++) fits = fits && lo[a] >= o[a] && hi[a] < o[a] + d[a];
        if (fits) return SPH_OK;
    }
    long long prod = 1;
    const int tdim[3] = {SPH_TX, SPH_TY, SPH_TZ};
    for (int a = 0; a < 3; a++) {
        origin[a] = lo[a] - 2;
        long long d = (long long)hi[a] - lo[a] + 1 + 4;
        d = (d + tdim[a] - 1) / tdim[a] * tdim[a];
        dims[a] = (int)std::min<long long>(std::max<long long>(d, 4), 1 << 20);
        prod *= dims[a];
    }
    while (prod > c->maxCells) {
        int a = (dims[0] >= dims[1] && dims[0] >= dims[2]) ? 0 : (dims[1] >= dims[2] ? 1 : 2);
        if (dims[a] <= 4) break;
        prod /= dims[a];
        dims[a] = std::max(4, (dims[a] / 2 + tdim[a] - 1) / tdim[a] * tdim[a]);
        prod *= dims[a];
    }
    set_grid(c, origin, dims, cellSize);
    return SPH_OK;
}

// ---- radix sort driver --------------------------------------------------------------------
// sorts (keysA, implicit index) by the low `bits` bits; result pointers returned
int radix_sort(SphCtx *c, uint32_t *keysA, uint32_t *valsA, uint32_t *keysB, uint32_t *valsB,
               int n, int bits, uint32_t **outKeys, uint32_t **outVals)
{
    const int nblocks = std::max(1, blocks_for(n, RS_TILE));
    int rc = ensure(c, c->hist, (size_t)256 * nblocks);
    if (rc) return rc;
    rc = ensure(c, c->rowTotal, 256, false);
    if (rc) return rc;
    const int passes = std::max(1, (bits + 7) / 8);
    uint32_t *kin = keysA, *vin = nullptr, *kout = keysB, *vout = valsB;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        k_radix_hist<<<nblocks, RS_THREADS, 0, c->stream>>>(kin, n, shift, c->hist.p, nblocks);
        k_radix_scan<<<256, RS_THREADS, 0, c->stream>>>(c->hist.p, c->rowTotal.p, nblocks);
        k_radix_scatter<<<nblocks, RS_THREADS, 0, c->stream>>>(kin, vin, kout, vout, n, shift,
                                                                c->hist.p, c->rowTotal.p,
                                                                nblocks);
        if (p == 0) {
            kin = keysB; vin = valsB; kout = keysA; vout = valsA;
        } else {
            std::swap(kin, kout);
            std::swap(vin, vout);
        }
    }
    HIPCHK(c, hipGetLastError());
    *outKeys = kin;
    *outVals = vin;
    return SPH_OK;
}

int ensure_fluid(SphCtx *c, int n)
{
    int rc = 0;
    const size_t N = (size_t)std::max(n, 1);
    for (int i = 0; i < 2 && !rc; i++) {
        rc = ensure(c, c->posm[i], N);
        if (!rc) rc = ensure(c, c->velq[i], N);
        if (!rc) rc = ensure(c, c->mat[i], N);
        if (!rc) rc = ensure(c, c->oid[i], N);
    }
    if (!rc) rc = ensure(c, c->keysA, N);
    if (!rc) rc = ensure(c, c->keysB, N);
    if (!rc) rc = ensure(c, c->valsA, N);
    if (!rc) rc = ensure(c, c->valsB, N);
    if (!rc) rc = ensure(c, c->rho, N);
    if (!rc) rc = ensure(c, c->prs, N);
    if (!rc) rc = ensure(c, c->force, N);
    if (!rc && !c->zeros.p) {
        rc = ensure(c, c->zeros, 16, false);
        if (!rc) HIPCHK(c, hipMemsetAsync(c->zeros.p, 0, 16 * sizeof(float4), c->stream));
    }
    return rc;
}

// grows the CURRENT fluid arrays to hold n records, keeping their first `keep`
template <typename T> int grow_keep(SphCtx *c, DevBuf<T> &b, size_t n, size_t keep)
{
    if (n <= b.cap && b.p) return SPH_OK;
    DevBuf<T> nb;
    int rc = ensure(c, nb, std::max(n, b.cap + b.cap / 2), false);
    if (rc) return rc;
    if (keep && b.p)
        HIPCHK(c, hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    release(b);
    b = nb;
    return SPH_OK;
}

int ensure_debug(SphCtx *c)
{
    int rc = 0;
    const size_t N = (size_t)std::max(c->n, 1);
    if (c->debug & SPH_DEBUG_HASHES) {
        rc = ensure(c, c->hashes, N);
        if (!rc) rc = ensure(c, c->keysEcs, N);
        if (!rc) rc = ensure(c, c->cells, 3 * N);
    }
    if (!rc && (c->debug & SPH_DEBUG_COUNTS)) {
        rc = ensure(c, c->dcount, N);
        if (!rc) rc = ensure(c, c->fcount, N);
    }
    return rc;
}

// boundary: keys -> sort -> table -> gather, once per (grid, boundary) change
int prepare_boundary(SphCtx *c)
{
    if (!c->boundaryDirty) return SPH_OK;
    int rc = ensure(c, c->btable, (size_t)c->grid.cells);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->btable.p, 0, (size_t)c->grid.cells * sizeof(int2), c->stream));
    if (c->nb > 0) {
        const int B = c->nb;
        rc = ensure(c, c->bposm, (size_t)B);
        if (!rc) rc = ensure(c, c->bkeysA, (size_t)B);
        if (!rc) rc = ensure(c, c->bkeysB, (size_t)B);
        if (!rc) rc = ensure(c, c->bvalsA, (size_t)B);
        if (!rc) rc = ensure(c, c->bvalsB, (size_t)B);
        if (rc) return rc;
        k_keys<false><<<blocks_for(B, 256), 256, 0, c->stream>>>(
            c->bposmIn.p, nullptr, B, c->grid, c->bkeysA.p, nullptr, nullptr, nullptr);
        uint32_t *sk, *sv;
        rc = radix_sort(c, c->bkeysA.p, c->bvalsA.p, c->bkeysB.p, c->bvalsB.p, B,
                        key_bits(c->grid.cells), &sk, &sv);
        if (rc) return rc;
        k_map_table<<<blocks_for(B, 256), 256, 0, c->stream>>>(sk, B, c->btable.p, nullptr);
        k_gather4<<<blocks_for(B, 256), 256, 0, c->stream>>>(sv, B, c->bposmIn.p, c->bposm.p);
        c->bperm = sv;
        HIPCHK(c, hipGetLastError());
    }
    c->boundaryDirty = false;
    return SPH_OK;
}

// HIP's current device is thread-local: every entry point that may allocate, launch or copy makes
// the context's device current first (a host may call from any thread)
int use_device(SphCtx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return SPH_OK;
}

int check_params(SphCtx *c, const SphParams *p)
{
    if (!c) return SPH_E_INVALID;
    if (int rc = use_device(c)) return rc;
    if (!p) return fail(c, SPH_E_INVALID, "params is NULL");
    if (!(p->gridSize > 0.f) || !std::isfinite(p->gridSize))
        return fail(c, SPH_E_INVALID, "gridSize must be positive and finite");
    if (!(p->kernelRadiusRate > 0.f)) return fail(c, SPH_E_INVALID, "kernelRadiusRate must be > 0");
    return SPH_OK;
}

// (re)derive everything that depends on the per-step scalars
int sync_params(SphCtx *c, const SphParams *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (c->n <= 0) return fail(c, SPH_E_INVALID, "no particles uploaded");
    if (c->matRate != p->kernelRadiusRate) {
        rc = upload_mats(c, p->kernelRadiusRate);
        if (rc) return rc;
    }
    if (!c->gridValid || c->grid.cellSize != p->gridSize) {
        if (c->gridExplicit) {
            const int o[3] = {c->grid.ox, c->grid.oy, c->grid.oz};
            const int d[3] = {c->grid.gx, c->grid.gy, c->grid.gz};
            set_grid(c, o, d, p->gridSize);
        } else {
            rc = auto_grid(c, p->gridSize);
            if (rc) return rc;
        }
        c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    }
    return SPH_OK;
}

} // namespace

// =============================================================================================
extern "C" {

int sph_create(const SphConfig *cfg, SphCtx **out)
{
    if (!out) return fail(nullptr, SPH_E_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SPH_E_NODEVICE, "no HIP device visible (libsphfluid has no CPU path)");
    const int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= ndev)
        return fail(nullptr, SPH_E_INVALID, "device %d out of range (%d visible)", dev, ndev);
    SphCtx *c = new (std::nothrow) SphCtx();
    if (!c) return fail(nullptr, SPH_E_NOMEM, "host allocation failed");
    c->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&c->prop, dev) != hipSuccess) {
        delete c;
        return fail(nullptr, SPH_E_DEVICE, "cannot open device %d", dev);
    }
    if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(nullptr, SPH_E_NODEVICE, "device %d is %s; this library is built for gfx950 only",
                    dev, arch.c_str());
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, SPH_E_DEVICE, "hipStreamCreate failed");
    }
    // optional: without them the plan simply runs on the main stream
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) c->stream2 = nullptr;
    if (c->stream2 && (hipEventCreateWithFlags(&c->evTable, hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&c->evPlan, hipEventDisableTiming) != hipSuccess)) {
        (void)hipStreamDestroy(c->stream2);
        c->stream2 = nullptr;
        if (c->evTable) (void)hipEventDestroy(c->evTable);
        if (c->evPlan) (void)hipEventDestroy(c->evPlan);
        c->evTable = c->evPlan = nullptr;
    }
    if (dev < SPH_MAX_DEVICES) { // this device's copy of `scratch` (device `dev` is current)
        std::lock_guard<std::mutex> lk(g_scratchLock);
        if (!g_scratchPtr[dev]) {
            void *ptr = nullptr;
            const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) {
                (void)hipStreamDestroy(c->stream);
                if (c->stream2) (void)hipStreamDestroy(c->stream2);
                if (c->evTable) (void)hipEventDestroy(c->evTable);
                if (c->evPlan) (void)hipEventDestroy(c->evPlan);
                delete c;
                return fail(nullptr, SPH_E_DEVICE, "hipGetSymbolAddress(scratch): %s",
                            hipGetErrorString(e));
            }
            g_scratchPtr[dev] = (SortScratch *)ptr;
        }
    }
    if (cfg && cfg->maxCells > 0) c->maxCells = std::max(cfg->maxCells, 64);
    c->timing = cfg ? std::min(std::max(cfg->enableTimings, 0), 2) : 0;
    c->pinHost = cfg && cfg->pinHostArrays;
    if (const char *e = getenv("SPHFLUID_POOL_LIMIT")) c->poolLimit = atoll(e);
    if (const char *e = getenv("SPHFLUID_STEP_SORT")) c->radixInSteps = strcmp(e, "radix") == 0;
    if (const char *e = getenv("SPHFLUID_SLAB_COUNT_PASS")) c->slabCountPass = atoi(e) != 0;
    if (const char *e = getenv("SPHFLUID_CAP1")) c->cap1 = std::max(0, std::min(NB_CAP, atoi(e)));
    *out = c;
    return SPH_OK;
}

int sph_destroy(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
#if defined(NB_STAMPS) || defined(NB_PLAN_TRACE)
    {
        unsigned long long h[32];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)) == hipSuccess) {
            const char *pn[8] = {"tile-start (wait for look-ups, LDS store)", "prefetch issue + next tile index", "records", "cells+windows", "scans+alloc", "hinfo+units", "slots+list", "end fence"};
            const char *en[8] = {"plan-dma", "rec-issue", "rec-wait", "units", "units-wait", "-", "-", "-"};
            for (int k = 0; k < 3; k++) {
                double tot = 0;
                for (int i = 0; i < 8; i++) tot += (double)h[8 * k + i];
                fprintf(stderr, "%s (%.3g Mcycles):", k == 0 ? "plan" : k == 1 ? "density" : "force", tot / 1e6);
                for (int i = 0; i < 8; i++)
                    if (h[8 * k + i]) fprintf(stderr, " %s %.1f%%", (k ? en : pn)[i], 100.0 * h[8 * k + i] / (tot > 0 ? tot : 1));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    for (int i = 0; i < 2; i++) {
        release(c->posm[i]); release(c->velq[i]); release(c->mat[i]); release(c->oid[i]);
    }
    release(c->keysA); release(c->keysB); release(c->valsA); release(c->valsB);
    release(c->rho); release(c->prs); release(c->force); release(c->zeros); release(c->dcount); release(c->fcount);
    release(c->hashes); release(c->keysEcs); release(c->cells);
    release(c->io3a); release(c->io3b); release(c->io1); release(c->ioU);
    release(c->dmats); release(c->table); release(c->btable); release(c->ttable); release(c->tileList); release(c->tileCtl); release(c->blockList); release(c->gcellList); release(c->gmaskOff); release(c->cellCounts); release(c->cellSums); release(c->pool);
    release(c->bpos3); release(c->bpar3); release(c->bposmIn); release(c->bposm);
    release(c->bkeysA); release(c->bkeysB); release(c->bvalsA); release(c->bvalsB);
    release(c->boxState); release(c->devBoxes); release(c->bodyImpulse); release(c->boxAabb); release(c->chunkAabb); release(c->bodyVelA); release(c->bodyVelB); release(c->bodyNc);
    release(c->hist); release(c->rowTotal); release(c->partial);
    for (const EventPair &ep : c->pending) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (hipEvent_t e : c->freeEvents) (void)hipEventDestroy(e);
    (void)sph_slab_comm_destroy(c);
    for (auto &e : c->pinned)
        if (e.bytes) (void)hipHostUnregister(const_cast<void *>(e.p));
    delete c->stage; // (after the stream has drained: joins its threads, frees the pinned ring)
    (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->evTable) (void)hipEventDestroy(c->evTable);
    if (c->evPlan) (void)hipEventDestroy(c->evPlan);
    delete c;
    return SPH_OK;
}

const char *sph_last_error(const SphCtx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int sph_device_name(SphCtx *c, char *name, size_t cap, int32_t *cus)
{
    if (!c) return SPH_E_INVALID;
    if (name && cap) snprintf(name, cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    if (cus) *cus = c->prop.multiProcessorCount;
    return SPH_OK;
}

int sph_set_debug(SphCtx *c, int32_t flags)
{
    if (!c) return SPH_E_INVALID;
    c->debug = flags;
    c->fallbackCap = (flags & SPH_DEBUG_NO_LDS_TILE) ? 0 : -1;
    c->havePlan = c->haveMasks = false;
    return SPH_OK;
}

// sticky device-side error word of the plan kernel (pool / list exhausted: sized never to happen)
static int check_plan_error(SphCtx *c)
{
    if (!c->tileCtl.p) return SPH_OK;
    uint32_t e = 0;
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    HIPCHK(c, hipMemcpyAsync(&e, c->tileCtl.p + NB_CTL_ERROR, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (e) HIPCHK(c, hipMemsetAsync(c->tileCtl.p + NB_CTL_ERROR, 0, 4, c->stream)); // reported once
    if (e) return fail(c, SPH_E_TOOLARGE, "neighbour plan overflow (flags %u): results of the last step are incomplete", e);
    return SPH_OK;
}

int sph_synchronize(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_plan_error(c);
}

float sph_find_grid_size(const SphParticle *params, const int32_t *chunk_first, int32_t nchunks,
                         float rate)
{
    // FindGridSizeJob.Execute, FluidSPHUtils.cs:35-43, gridSize[0] starts at float.MaxValue
    float grid = 3.402823466e+38f;
    for (int c = 0; c < nchunks; c++) {
        const float radius = params[chunk_first[c]].radius;
        if (radius < grid / rate) grid = radius * rate;
    }
    return grid;
}

int sph_set_boundary(SphCtx *c, const float *bpos, const SphBoundaryParticle *bparams, int32_t B)
{
    if (!c) return SPH_E_INVALID;
    if (B < 0 || (B > 0 && (!bpos || !bparams))) return fail(c, SPH_E_INVALID, "bad boundary arrays");
    HIPCHK(c, hipSetDevice(c->device));
    c->nb = B;
    c->boundaryDirty = true;
    if (B == 0) return SPH_OK;
    int rc = ensure(c, c->bpos3, (size_t)3 * B);
    if (!rc) rc = ensure(c, c->bpar3, (size_t)3 * B);
    if (!rc) rc = ensure(c, c->bposmIn, (size_t)B);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->bpos3.p, bpos, (size_t)12 * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->bpar3.p, bparams, (size_t)12 * B, hipMemcpyHostToDevice, c->stream));
    k_pack_boundary<<<blocks_for(B, 256), 256, 0, c->stream>>>(c->bpos3.p, c->bpar3.p, B,
                                                               c->bposmIn.p);
    for (int a = 0; a < 3; a++) {
        c->hostBMin[a] = 3.402823466e+38f;
        c->hostBMax[a] = -3.402823466e+38f;
    }
    for (int i = 0; i < B; i++)
        for (int a = 0; a < 3; a++) {
            c->hostBMin[a] = std::min(c->hostBMin[a], bpos[3 * i + a]);
            c->hostBMax[a] = std::max(c->hostBMax[a], bpos[3 * i + a]);
        }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->gridExplicit) c->gridValid = false; // re-derive the table around the new boundary
    return SPH_OK;
}

int sph_set_colliders(SphCtx *c, const SphBox *boxes, int32_t K)
{
    if (!c) return SPH_E_INVALID;
    if (K < 0 || (K > 0 && !boxes)) return fail(c, SPH_E_INVALID, "bad collider array");
    if (K > 65536) return fail(c, SPH_E_TOOLARGE, "at most 65536 box colliders");
    HIPCHK(c, hipSetDevice(c->device));
    c->nboxes = K;
    c->anyDynamicBox = false;
    for (int k = 0; k < K; k++)
        c->anyDynamicBox = c->anyDynamicBox || boxes[k].invMass != 0.f ||
                           boxes[k].invInertia[0] != 0.f || boxes[k].invInertia[1] != 0.f ||
                           boxes[k].invInertia[2] != 0.f;
    if (K == 0) return SPH_OK;
    int rc = ensure(c, c->boxState, (size_t)25 * K);
    if (!rc) rc = ensure(c, c->devBoxes, (size_t)K);
    if (!rc) rc = ensure(c, c->bodyImpulse, (size_t)6 * K);
    if (!rc) rc = ensure(c, c->boxAabb, (size_t)2 * K);
    if (!rc) rc = ensure(c, c->chunkAabb, (size_t)2 * blocks_for(K, SPH_BOX_CHUNK));
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->boxState.p, boxes, sizeof(SphBox) * (size_t)K,
                             hipMemcpyHostToDevice, c->stream));
    k_bodies<<<blocks_for(K, 64), 64, 0, c->stream>>>(c->boxState.p, c->devBoxes.p, c->boxAabb.p,
                                                      c->bodyImpulse.p, nullptr, K, 0, 0.f, 0.f, 0.f,
                                                      0.f, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    k_box_chunks<<<blocks_for(blocks_for(K, SPH_BOX_CHUNK), 64), 64, 0, c->stream>>>(
        c->boxAabb.p, K, c->chunkAabb.p);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_get_colliders(SphCtx *c, SphBox *out, int32_t K)
{
    if (!c || !out) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (K != c->nboxes) return fail(c, SPH_E_INVALID, "K=%d but %d colliders are set", K, c->nboxes);
    if (K == 0) return SPH_OK;
    HIPCHK(c, hipMemcpyAsync(out, c->boxState.p, sizeof(SphBox) * (size_t)K, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_set_grid(SphCtx *c, const SphGrid *g)
{
    if (!c) return SPH_E_INVALID;
    if (!g) { // back to automatic sizing
        c->gridExplicit = false;
        c->gridValid = false;
        return SPH_OK;
    }
    long long prod = 1;
    for (int a = 0; a < 3; a++) {
        if (g->dims[a] < 4) return fail(c, SPH_E_INVALID, "every grid dimension must be >= 4");
        prod *= g->dims[a];
    }
    if (prod > (1ll << 30)) return fail(c, SPH_E_TOOLARGE, "cell table beyond 2^30 cells");
    c->gridExplicit = true;
    set_grid(c, g->origin, g->dims, c->gridValid ? c->grid.cellSize : 0.f);
    c->gridValid = false; // cell size arrives with the first SphParams
    c->grid.gx = g->dims[0]; c->grid.gy = g->dims[1]; c->grid.gz = g->dims[2];
    c->grid.ox = g->origin[0]; c->grid.oy = g->origin[1]; c->grid.oz = g->origin[2];
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    return SPH_OK;
}

int sph_get_grid(SphCtx *c, SphGrid *g)
{
    if (!c || !g) return SPH_E_INVALID;
    if (!c->gridValid) return fail(c, SPH_E_INVALID, "grid not derived yet (run a stage first)");
    g->origin[0] = c->grid.ox; g->origin[1] = c->grid.oy; g->origin[2] = c->grid.oz;
    g->dims[0] = c->grid.gx; g->dims[1] = c->grid.gy; g->dims[2] = c->grid.gz;
    g->cells = c->grid.cells;
    g->keyBits = key_bits(c->grid.cells);
    return SPH_OK;
}

// Registers a caller array with the driver once (SphConfig.pinHostArrays) and remembers it; a
// failure is not an error -- the copy then runs from pageable memory as before -- but it is counted
// (sph_pin_stats).  An array that comes back with ANY other size is registered afresh: the host may
// have freed and reallocated it, and a smaller mapping at the same address has other pages.  At
// most 16 arrays are kept, the oldest is released first.
static void pin_host(SphCtx *c, const void *p, size_t bytes)
{
    if (!c->pinHost || !p || bytes < (1u << 20)) return;
    auto reg = [&](size_t n) -> size_t {
        const hipError_t e = hipHostRegister(const_cast<void *>(p), n, hipHostRegisterDefault);
        if (e == hipSuccess) {
            c->pinOk++;
            return n;
        }
        (void)hipGetLastError();
        c->pinFailed++;
        // not an error of the call in progress, but the text is kept where the host can read it
        char buf[256];
        snprintf(buf, sizeof(buf), "hipHostRegister(%zu bytes) failed: %s -- the array goes through the "
                                   "library's pinned staging ring instead", n, hipGetErrorString(e));
        c->pinErr = buf;
        return 0;
    };
    for (auto &e : c->pinned)
        if (e.p == p) {
            if (e.bytes == bytes || (e.bytes == 0 && e.asked == bytes)) return; // (a failure is not retried every frame)
            if (e.bytes) (void)hipHostUnregister(const_cast<void *>(p));
            e.bytes = reg(bytes);
            e.asked = bytes;
            return;
        }
    if (c->pinned.size() >= 16) {
        if (c->pinned.front().bytes) (void)hipHostUnregister(const_cast<void *>(c->pinned.front().p));
        c->pinned.erase(c->pinned.begin());
    }
    c->pinned.push_back({p, reg(bytes), bytes});
}

// Releases the registration of one caller array (p) or of all of them (p == NULL).  A host that
// frees or reallocates an array it has passed to a context with pinHostArrays set calls this first:
// freeing registered memory is undefined, and a new mapping at the same address has other pages.
int sph_unpin_host(SphCtx *c, const void *p)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    HIPCHK(c, hipStreamSynchronize(c->stream)); // no copy may still be using the pages
    for (size_t i = 0; i < c->pinned.size();) {
        if (!p || c->pinned[i].p == p) {
            if (c->pinned[i].bytes && hipHostUnregister(const_cast<void *>(c->pinned[i].p)) != hipSuccess)
                (void)hipGetLastError();
            c->pinned.erase(c->pinned.begin() + (long)i);
        } else {
            i++;
        }
    }
    return SPH_OK;
}

// stats[0] = arrays registered at the moment, [1] = registrations that succeeded, [2] = that failed
// (the copies of those arrays ran from pageable memory), since sph_create
int sph_pin_stats(SphCtx *c, int32_t stats[3])
{
    if (!c || !stats) return SPH_E_INVALID;
    int live = 0;
    for (const auto &e : c->pinned) live += e.bytes ? 1 : 0;
    stats[0] = live;
    stats[1] = c->pinOk;
    stats[2] = c->pinFailed;
    if (c->pinFailed && !c->pinErr.empty()) c->err = c->pinErr; // (sph_last_error gives the reason)
    return SPH_OK;
}

int sph_upload(SphCtx *c, const float *pos, const float *vel, const float *invMass,
               const SphParticle *params, int32_t N, float gridSize)
{
    if (!c) return SPH_E_INVALID;
    if (N <= 0 || !pos || !vel || !invMass || !params)
        return fail(c, SPH_E_INVALID, "sph_upload needs N > 0 and non-NULL arrays");
    if (!(gridSize > 0.f) || !std::isfinite(gridSize))
        return fail(c, SPH_E_INVALID, "gridSize must be positive and finite");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_fluid(c, N);
    if (!rc) rc = ensure(c, c->io3a, (size_t)3 * N);
    if (!rc) rc = ensure(c, c->io3b, (size_t)3 * N);
    if (!rc) rc = ensure(c, c->ioU, (size_t)N);
    if (rc) return rc;
    build_materials(c, invMass, params, N);
    rc = upload_mats(c, c->matRate > 0.f ? c->matRate : 4.0f);
    if (rc) return rc;
    c->n = N;
    c->cur = 0;
    c->slab = false;
    c->nOwned = c->nGhost = c->nDead = 0;
    pin_host(c, pos, (size_t)12 * N);
    pin_host(c, vel, (size_t)12 * N);
    if ((rc = copy_h2d(c, c->io3a.p, pos, (size_t)12 * N))) return rc;
    if ((rc = copy_h2d(c, c->io3b.p, vel, (size_t)12 * N))) return rc;
    if (c->matUniform)
        HIPCHK(c, hipMemsetAsync(c->ioU.p, 0, (size_t)4 * N, c->stream));
    else
        HIPCHK(c, hipMemcpyAsync(c->ioU.p, c->hostMatIdx.data(), (size_t)4 * N, hipMemcpyHostToDevice,
                                 c->stream));
    k_pack_fluid<<<blocks_for(N, 256), 256, 0, c->stream>>>(c->io3a.p, c->io3b.p, c->ioU.p,
                                                            c->dmats.p, N, c->posm[0].p,
                                                            c->velq[0].p, c->mat[0].p, c->oid[0].p);
    HIPCHK(c, hipGetLastError());
    // one material (the usual scene): the index array is all zeros and need not travel with the
    // particles -- 8 of the reorder pass's 96 bytes per particle, 4 of the integrate pass's 72
    c->matAllZero = false;
    if (c->matUniform) {
        HIPCHK(c, hipMemsetAsync(c->mat[1].p, 0, (size_t)4 * N, c->stream));
        c->matAllZero = true;
    }
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    if (c->gridExplicit) {
        const int o[3] = {c->grid.ox, c->grid.oy, c->grid.oz};
        const int d[3] = {c->grid.gx, c->grid.gy, c->grid.gz};
        set_grid(c, o, d, gridSize);
    } else {
        rc = auto_grid(c, gridSize);
        if (rc) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ensure_debug(c);
}

int sph_download(SphCtx *c, float *pos_out, float *vel_out)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (c->n <= 0) return fail(c, SPH_E_INVALID, "no particles uploaded");
    if (c->slab) return fail(c, SPH_E_INVALID, "slab mode: use sph_slab_download");
    const int N = c->n;
    int rc = ensure(c, c->io3a, (size_t)3 * N);
    if (!rc) rc = ensure(c, c->io3b, (size_t)3 * N);
    if (!rc) rc = check_plan_error(c);
    if (rc) return rc;
    const int cur = c->cur;
    pin_host(c, pos_out, (size_t)12 * N);
    pin_host(c, vel_out, (size_t)12 * N);
    // (both unpack kernels first: the second runs while the first array crosses the link)
    if (pos_out)
        k_unpack3<<<blocks_for(N, 256), 256, 0, c->stream>>>(c->posm[cur].p, c->oid[cur].p, N,
                                                             c->io3a.p);
    if (vel_out)
        k_unpack3<<<blocks_for(N, 256), 256, 0, c->stream>>>(c->velq[cur].p, c->oid[cur].p, N,
                                                             c->io3b.p);
    HIPCHK(c, hipGetLastError());
    if (pos_out && (rc = copy_d2h(c, pos_out, c->io3a.p, (size_t)12 * N))) return rc;
    if (vel_out && (rc = copy_d2h(c, vel_out, c->io3b.p, (size_t)12 * N))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

// ---- stages -------------------------------------------------------------------------------
int sph_hash_position(SphCtx *c, const SphParams *p)
{
    int rc = sync_params(c, p);
    if (rc) return rc;
    rc = ensure_debug(c);
    if (rc) return rc;
    if (c->slab && c->slabKeys) { // the pack and import kernels of this step's exchange have written them
        c->slabKeys = false;
        c->haveKeys = true;
        c->haveSort = c->haveTable = c->ordered = false;
        return SPH_OK;
    }
    StageTimer t(c, SPH_T_KEYS);
    const int N = c->n, cur = c->cur;
    if (c->slab)
        k_slab_keys<<<blocks_for(N, 256), 256, 0, c->stream>>>(c->posm[cur].p, c->oid[cur].p, N,
                                                               c->grid, c->keysA.p);
    else if (c->debug & SPH_DEBUG_HASHES)
        k_keys<true><<<blocks_for(N, 256), 256, 0, c->stream>>>(c->posm[cur].p, c->oid[cur].p, N,
                                                                c->grid, c->keysA.p, c->hashes.p,
                                                                c->cells.p, c->keysEcs.p);
    else
        k_keys<false><<<blocks_for(N, 256), 256, 0, c->stream>>>(c->posm[cur].p, nullptr, N,
                                                                 c->grid, c->keysA.p, nullptr,
                                                                 nullptr, nullptr);
    HIPCHK(c, hipGetLastError());
    c->haveKeys = true;
    c->haveSort = c->haveTable = c->ordered = false;
    return SPH_OK;
}

int sph_radix_sort(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (!c->haveKeys) return fail(c, SPH_E_INVALID, "sph_radix_sort before sph_hash_position");
    StageTimer t(c, SPH_T_SORT);
    // slab mode: dead ghosts carry key = cells, sort to the end and are cut off here
    int rc = radix_sort(c, c->keysA.p, c->valsA.p, c->keysB.p, c->valsB.p, c->n,
                        key_bits(c->grid.cells + (c->slab ? 1 : 0)), &c->sortedKeys,
                        &c->sortedVals);
    if (rc) return rc;
    if (c->slab) {
        c->n -= c->nDead;
        c->nDead = 0;
    }
    c->haveSort = true;
    c->haveKeys = false; // keysA may have been overwritten
    c->haveTable = c->ordered = false;
    return SPH_OK;
}

int sph_map_table(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (!c->haveSort) return fail(c, SPH_E_INVALID, "sph_map_table before sph_radix_sort");
    int rc = ensure(c, c->table, (size_t)c->grid.cells);
    if (!rc) rc = ensure(c, c->ttable, (size_t)c->grid.tiles);
    if (!rc) rc = ensure(c, c->tileList, (size_t)c->grid.tiles);
    if (!rc && !c->tileCtl.p) {
        rc = ensure(c, c->tileCtl, SPH_QUEUE_SLOTS + 1, false);
        if (!rc) HIPCHK(c, hipMemsetAsync(c->tileCtl.p, 0, (SPH_QUEUE_SLOTS + 1) * sizeof(uint32_t), c->stream));
    }
    if (rc) return rc;
    rc = prepare_boundary(c);
    if (rc) return rc;
    StageTimer t(c, SPH_T_TABLE);
    HIPCHK(c, hipMemsetAsync(c->table.p, 0, (size_t)c->grid.cells * sizeof(int2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->ttable.p, 0, (size_t)c->grid.tiles * sizeof(int2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->tileCtl.p, 0, SPH_QUEUE_SLOTS * sizeof(uint32_t), c->stream));
    c->queueNext = NB_CTL_QUEUE0;
    c->havePlan = c->haveMasks = false;
    k_map_table<<<blocks_for(c->n, 256), 256, 0, c->stream>>>(c->sortedKeys, c->n, c->table.p,
                                                              c->ttable.p);
    k_tile_list<<<blocks_for(c->grid.tiles, 256 * TL_ITEMS), 256, 0, c->stream>>>(
        c->ttable.p, c->grid.tiles, c->tileList.p, c->tileCtl.p);
    HIPCHK(c, hipGetLastError());
    c->haveTable = true;
    c->ordered = false;
    return SPH_OK;
}

// Sort + tables of the step entry points: the cell counting sort of sph_sort.h (the stage calls
// above keep the radix sort and its stable permutation).  Leaves what sph_radix_sort and
// sph_map_table leave: sortedKeys / sortedVals (a sorting permutation; the order inside a cell is
// arbitrary, sph_reorder ranks every cell by particle id), table, ttable, tileList.
static int sort_and_table(SphCtx *c)
{
    int rc;
    if (c->radixInSteps) {
        if ((rc = sph_radix_sort(c))) return rc;
        return sph_map_table(c);
    }
    if (!c->haveKeys) return fail(c, SPH_E_INVALID, "sort before sph_hash_position");
    const int n = c->n, cells = c->grid.cells;
    const int nquads = (cells + 1 + 3) / 4; // (+ 1: slab mode's dead ghosts carry key = cells)
    const int nb = blocks_for(nquads, 256);
    rc = ensure(c, c->cellCounts, (size_t)nquads * 4);
    if (!rc) rc = ensure(c, c->cellSums, (size_t)nb);
    if (!rc) rc = ensure(c, c->table, (size_t)cells);
    if (!rc) rc = ensure(c, c->ttable, (size_t)c->grid.tiles);
    if (!rc) rc = ensure(c, c->tileList, (size_t)c->grid.tiles);
    if (!rc && !c->tileCtl.p) {
        rc = ensure(c, c->tileCtl, SPH_QUEUE_SLOTS + 1, false);
        if (!rc) HIPCHK(c, hipMemsetAsync(c->tileCtl.p, 0, (SPH_QUEUE_SLOTS + 1) * sizeof(uint32_t), c->stream));
    }
    if (rc) return rc;
    rc = prepare_boundary(c);
    if (rc) return rc;
    {
        StageTimer t(c, SPH_T_SORT);
        HIPCHK(c, hipMemsetAsync(c->cellCounts.p, 0, (size_t)nquads * 16, c->stream));
        k_cell_count<<<blocks_for(n, 256 * CS_COUNT_ITEMS), 256, 0, c->stream>>>(c->keysA.p, n, c->cellCounts.p, c->valsA.p);
    }
    {
        StageTimer t(c, SPH_T_TABLE);
        HIPCHK(c, hipMemsetAsync(c->tileCtl.p, 0, SPH_QUEUE_SLOTS * sizeof(uint32_t), c->stream));
        c->queueNext = NB_CTL_QUEUE0;
        c->havePlan = c->haveMasks = false;
        k_cell_sums<<<nb, 256, 0, c->stream>>>((const uint4 *)c->cellCounts.p, nquads, c->cellSums.p);
        k_cell_scan_sums<<<1, RS_THREADS, 0, c->stream>>>(c->cellSums.p, nb);
        k_cell_table<<<nb, 256, 0, c->stream>>>((const uint4 *)c->cellCounts.p, nquads, c->cellSums.p, cells,
                                                c->table.p, c->ttable.p);
        k_tile_list<<<blocks_for(c->grid.tiles, 256 * TL_ITEMS), 256, 0, c->stream>>>(
            c->ttable.p, c->grid.tiles, c->tileList.p, c->tileCtl.p);
    }
    {
        StageTimer t(c, SPH_T_SORT);
        k_cell_place<<<blocks_for(n, 256 * CS_PLACE_ITEMS), 256, 0, c->stream>>>(
            c->keysA.p, c->valsA.p, n, c->table.p, (uint32_t)cells, (uint32_t)(n - (c->slab ? c->nDead : 0)),
            c->valsB.p, c->keysB.p);
    }
    HIPCHK(c, hipGetLastError());
    c->sortedKeys = c->keysB.p;
    c->sortedVals = c->valsB.p;
    if (c->slab) { // dead ghosts carry key = cells, sort to the end and are cut off here
        c->n -= c->nDead;
        c->nDead = 0;
    }
    c->haveSort = c->haveTable = true;
    c->haveKeys = false;
    c->ordered = false;
    return SPH_OK;
}

int sph_reorder(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (!c->haveTable) return fail(c, SPH_E_INVALID, "sph_reorder before sph_map_table");
    if (c->ordered) return SPH_OK;
    StageTimer t(c, SPH_T_REORDER);
    const int N = c->n, cur = c->cur, nxt = cur ^ 1;
    k_reorder<<<blocks_for(N, 256), 256, 0, c->stream>>>(
        c->sortedVals, c->sortedKeys, c->table.p, N, c->posm[cur].p, c->velq[cur].p,
        c->matAllZero ? nullptr : c->mat[cur].p, c->oid[cur].p, c->posm[nxt].p, c->velq[nxt].p,
        c->mat[nxt].p, c->oid[nxt].p);
    HIPCHK(c, hipGetLastError());
    c->cur = nxt;
    c->ordered = true;
    c->haveMasks = false; // (the plan depends on the table only: it stays)
    c->haveDensity = c->haveForce = false;
    c->rhoInOrder = false;
    return SPH_OK;
}

static NbArgs neighbor_args(SphCtx *c, const SphParams *p, int cap)
{
    const int cur = c->cur;
    NbArgs A{};
    A.posm = c->posm[cur].p;
    A.velq = c->velq[cur].p;
    A.mat = c->mat[cur].p;
    A.rhoIn = c->rho.p;
    A.prsIn = c->prs.p;
    A.table = c->table.p;
    A.ttable = c->ttable.p;
    A.btable = c->nb > 0 ? c->btable.p : nullptr;
    A.bposm = c->nb > 0 ? c->bposm.p : nullptr;
    A.mats = c->dmats.p;
    A.zeros = c->zeros.p;
    A.rho = c->rho.p;
    A.prs = c->prs.p;
    A.force = c->force.p;
    A.counts = nullptr;
    A.g = c->grid;
    A.gx = p->gravity[0];
    A.gy = p->gravity[1];
    A.gz = p->gravity[2];
    A.cap = c->fallbackCap >= 0 ? std::min(c->fallbackCap, cap) : cap;
    A.cap1 = std::min(A.cap, c->cap1);
    // window tails may read on into the next cells of the row only if no support radius exceeds
    // the cell size (then those records are farther than h from every target of the cell)
    A.overread = 1;
    for (const Mat &m : c->mats)
        if (!(m.h <= c->grid.cellSize)) A.overread = 0;
    A.nmats = (int)c->mats.size();
    A.tileList = c->tileList.p;
    A.ctl = c->tileCtl.p;
    A.blockList = c->blockList.p;
    A.blockCap = (int)std::min<size_t>(c->blockList.cap, 1u << 30);
    A.gcellList = c->gcellList.p;
    A.gcellCap = (int)std::min<size_t>(std::min(c->gcellList.cap, c->gmaskOff.cap), 1u << 30);
    A.gmaskOff = c->gmaskOff.p;
    A.planGrid = (uint32_t)c->planGrid;
    A.pool = c->pool.p;
    A.poolCap = (uint32_t)std::min<size_t>(c->pool.cap, 0xffffffffu);
    return A;
}

// The plan of this ordering (k_plan): once per step, used by the density and the force launch.
static int ensure_plan(SphCtx *c, const SphParams *p, hipStream_t stream = nullptr)
{
    if (c->havePlan) return SPH_OK;
    if (!stream) stream = c->stream;
    if (c->mats.size() > 65535) return fail(c, SPH_E_TOOLARGE, "more than 65535 distinct materials");
    // Sizes.  Block list: a block has at least one target and a tile at most 32 blocks.  Global-
    // memory path: a cell takes it only if more than NB_CAP records lie in its 27 cells.  Pool, in
    // bytes: per non-empty tile the halo table (2304 B for a whole-tile block; a tile that falls
    // into single-cell blocks, 432 B each, holds hundreds of particles and is covered by their
    // share); per non-empty cell one unit (32 B), one hit-mask word for its 64 lanes (256 B) and
    // nine windows' slots -- a lattice with one particle per cell needs exactly that, 330 B per
    // particle; per particle beyond that ~37 B of masks (one bit per lane and slot) and ~14 B of
    // slots at eight per cell, more in the sparse cells the per-cell term already pays for.
    const size_t N = (size_t)c->n, tiles = (size_t)c->grid.tiles;
    const size_t tilesBound = std::min(tiles, N), cellsBound = std::min((size_t)c->grid.cells, N);
    // persistent single-wave workgroups (each takes its pieces of the pool and of the list)
    const int planWgs = NB_PLAN_WGS_PER_CU * std::max(1, c->prop.multiProcessorCount);
    const int grid = std::max(1, std::min((int)std::min<size_t>(tilesBound, 1u << 30), planWgs));
    // (+ two entries per tile: a plan workgroup's runs of the list have one entry per tile it has
    // still to plan, and a tile without fluid targets leaves its entry unused)
    const BlockHdr *listWas = c->blockList.p;
    int rc = ensure(c, c->blockList, std::min(tilesBound * 32, N) + 2 * tilesBound + 64 + (size_t)grid * (NB_LIST_PIECE + 2));
    // (an entry the plan could not fill is written as empty by the plan itself; the rest of a fresh
    // list must not look like blocks either)
    if (!rc && c->blockList.p != listWas)
        HIPCHK(c, hipMemsetAsync(c->blockList.p, 0, c->blockList.cap * sizeof(BlockHdr), stream));
    // (+ a run of entries per plan workgroup: they are taken NB_GRUN at a time)
    const size_t gcells = (c->fallbackCap >= 0 || c->cap1 < NB_CAP ? N : (N + (size_t)c->nb) / 40) + 64 + (size_t)grid * NB_GRUN;
    if (!rc) rc = ensure(c, c->gcellList, gcells);
    if (!rc) rc = ensure(c, c->gmaskOff, gcells);
    size_t poolUnits = (tilesBound * 2560 + cellsBound * 352 + N * 128) / 16 + 1024 + (size_t)grid * NB_POOL_PIECE;
    // (SPH_DEBUG_NO_LDS_TILE sends every cell down the crowded-cell path, whose masks are per group
    // of targets: a word of 64 lanes per target of a sparse cell)
    if (c->fallbackCap >= 0) poolUnits += N * 512 / 16;
    if (c->poolLimit > 0) poolUnits = std::min<size_t>(poolUnits, (size_t)c->poolLimit); // (tests)
    if (!rc) rc = ensure(c, c->pool, poolUnits, c->poolLimit <= 0);
    if (rc) return rc;
    StageTimer t(c, SPH_T_PLAN, stream);
    c->planGrid = grid;
    NbArgs A = neighbor_args(c, p, NB_CAP);
    if (c->poolLimit > 0) A.poolCap = std::min<uint32_t>(A.poolCap, (uint32_t)c->poolLimit);
    k_plan<<<grid, NB_PTHREADS, 0, stream>>>(A);
    HIPCHK(c, hipGetLastError());
    c->havePlan = true;
    return SPH_OK;
}

// launches k_neighbors<FORCE,COUNT> as persistent workgroups: as many as the chip holds at once
extern "C++" {
template <bool FORCE, bool COUNT, bool MULTI> static int launch_neighbors_m(SphCtx *c, NbArgs &A)
{
    constexpr int threads = FORCE ? NB_FTHREADS : NB_DTHREADS;
    int &grid = c->nbGrid[(FORCE ? 2 : 0) + (COUNT ? 1 : 0) + (MULTI ? 4 : 0)];
    if (grid == 0) {
        int perCU = 0;
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, k_neighbors<FORCE, COUNT, MULTI>,
                                                               threads, 0));
        grid = std::max(1, perCU) * std::max(1, c->prop.multiProcessorCount);
    }
    // every launch takes a fresh (zero) queue slot; they are all cleared with the tile table, so
    // there is no per-launch memset (a 4-byte hipMemsetAsync costs 20 us on the device)
    if (c->queueNext + 2 > SPH_QUEUE_SLOTS) {
        HIPCHK(c, hipMemsetAsync(c->tileCtl.p + NB_CTL_QUEUE0, 0,
                                 (SPH_QUEUE_SLOTS - NB_CTL_QUEUE0) * sizeof(uint32_t), c->stream));
        c->queueNext = NB_CTL_QUEUE0;
    }
    A.queue = c->queueNext;
    c->queueNext += 2; // blocks, crowded cells
    k_neighbors<FORCE, COUNT, MULTI><<<std::min(grid, std::max(c->grid.tiles, 1)), threads, 0, c->stream>>>(A);
    HIPCHK(c, hipGetLastError());
    return SPH_OK;
}
// (one instantiation for scenes with a single material, one for the others: see exec_block)
template <bool FORCE, bool COUNT> static int launch_neighbors(SphCtx *c, NbArgs &A)
{
    return A.nmats > 1 ? launch_neighbors_m<FORCE, COUNT, true>(c, A) : launch_neighbors_m<FORCE, COUNT, false>(c, A);
}
} // extern "C++"

int sph_compute_density_and_pressure(SphCtx *c, const SphParams *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (!c->ordered) return fail(c, SPH_E_INVALID, "density before sph_reorder");
    rc = ensure_debug(c);
    if (rc) return rc;
    if ((rc = ensure_plan(c, p))) return rc;
    StageTimer t(c, SPH_T_DENSITY);
    NbArgs A = neighbor_args(c, p, NB_CAP);
    if (c->debug & SPH_DEBUG_COUNTS) {
        A.counts = c->dcount.p;
        rc = launch_neighbors<false, true>(c, A);
    } else {
        rc = launch_neighbors<false, false>(c, A);
        if (!rc) c->haveMasks = true;
    }
    if (rc) return rc;
    c->haveDensity = true;
    c->rhoInOrder = true;
    return SPH_OK;
}

int sph_compute_force(SphCtx *c, const SphParams *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (!c->haveDensity) return fail(c, SPH_E_INVALID, "force before density");
    if ((rc = ensure_plan(c, p))) return rc;
    StageTimer t(c, SPH_T_FORCE);
    NbArgs A = neighbor_args(c, p, NB_CAP);
    if (c->debug & SPH_DEBUG_COUNTS) {
        A.counts = c->fcount.p;
        rc = launch_neighbors<true, true>(c, A);
    } else {
        A.useMasks = c->haveMasks ? 1 : 0;
        rc = launch_neighbors<true, false>(c, A);
    }
    if (rc) return rc;
    c->haveForce = true;
    return SPH_OK;
}

static int launch_apply_integrate(SphCtx *c, const SphParams *p, bool apply, bool integrate,
                                  bool nextKeys = false)
{
    const int N = c->n, cur = c->cur;
    const int nb = blocks_for(N, 256);
    if (integrate) c->slabKeys = false; // (the positions change)
    uint32_t *slabNext = nullptr;
    c->nextCounts = false;
    if (integrate && c->slab && c->expectHalo >= 0.f && c->slabNext.p) {
        HIPCHK(c, hipMemsetAsync(c->slabNext.p, 0, 16, c->stream));
        slabNext = c->slabNext.p;
        c->nextCounts = true;
    }
#define LAUNCH_AI(A, I, S)                                                                       \
    k_apply_integrate<A, I, S><<<nb, 256, 0, c->stream>>>(                                          \
        c->posm[cur].p, c->velq[cur].p, c->force.p, c->rho.p,                                    \
        c->matAllZero ? nullptr : c->mat[cur].p, c->dmats.p, N,                                  \
        p->dt, c->devBoxes.p, c->nboxes, p->particleRestitution, c->bodyImpulse.p,               \
        c->boxAabb.p, c->chunkAabb.p,                                                            \
        c->slab ? c->oid[cur].p : nullptr, nextKeys ? c->keysA.p : nullptr,                      \
        slabNext, c->expectLo, c->expectHi, c->expectHalo, c->grid)
    if (apply && integrate && slabNext) LAUNCH_AI(true, true, true);
    else if (apply && integrate) LAUNCH_AI(true, true, false);
    else if (apply) LAUNCH_AI(true, false, false);
    else if (slabNext) LAUNCH_AI(false, true, true);
    else LAUNCH_AI(false, true, false);
#undef LAUNCH_AI
    HIPCHK(c, hipGetLastError());
    return SPH_OK;
}

static int launch_bodies(SphCtx *c, const SphParams *p)
{
    if (c->nboxes <= 0) return SPH_OK;
    StageTimer t(c, SPH_T_BODIES);
    const int K = c->nboxes, nb = blocks_for(K, 64);
#define LAUNCH_BODIES(MODE, VELIN)                                                               \
    k_bodies<<<nb, 64, 0, c->stream>>>(c->boxState.p, c->devBoxes.p, c->boxAabb.p,               \
                                       c->bodyImpulse.p, VELIN, K, MODE, p->dt, p->gravity[0],   \
                                       p->gravity[1], p->gravity[2], p->boundBodies,             \
                                       p->bodyMin[0], p->bodyMin[1], p->bodyMin[2],              \
                                       p->bodyMax[0], p->bodyMax[1], p->bodyMax[2])
    if (p->bodyContacts && c->anyDynamicBox && K >= 2) {
        // velocities -> contacts between boxes (count, four Jacobi passes) -> tank contacts, pose
        int rc = ensure(c, c->bodyVelA, (size_t)6 * K);
        if (!rc) rc = ensure(c, c->bodyVelB, (size_t)6 * K);
        if (!rc) rc = ensure(c, c->bodyNc, (size_t)K);
        if (rc) return rc;
        LAUNCH_BODIES(2, nullptr);
        k_body_contact_count<<<nb, 64, 0, c->stream>>>(c->boxState.p, c->boxAabb.p, K, c->bodyNc.p,
                                                       c->bodyVelA.p);
        for (int it = 0; it < 4; it++)
            k_body_contact_pass<<<nb, 64, 0, c->stream>>>(
                c->boxState.p, c->boxAabb.p, c->bodyNc.p, K, p->dt,
                (it & 1) ? c->bodyVelB.p : c->bodyVelA.p, (it & 1) ? c->bodyVelA.p : c->bodyVelB.p);
        LAUNCH_BODIES(3, c->bodyVelA.p);
    } else {
        LAUNCH_BODIES(1, nullptr);
    }
#undef LAUNCH_BODIES
    if (c->anyDynamicBox) // (static boxes keep the chunk bounds of sph_set_colliders)
        k_box_chunks<<<blocks_for(blocks_for(K, SPH_BOX_CHUNK), 64), 64, 0, c->stream>>>(
            c->boxAabb.p, K, c->chunkAabb.p);
    HIPCHK(c, hipGetLastError());
    return SPH_OK;
}

int sph_apply_force(SphCtx *c, const SphParams *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (!c->haveForce) return fail(c, SPH_E_INVALID, "apply_force before compute_force");
    StageTimer t(c, SPH_T_INTEGRATE);
    return launch_apply_integrate(c, p, true, false);
}

int sph_integrate(SphCtx *c, const SphParams *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (c->n <= 0) return fail(c, SPH_E_INVALID, "no particles uploaded");
    {
        StageTimer t(c, SPH_T_INTEGRATE);
        rc = launch_apply_integrate(c, p, false, true);
        if (rc) return rc;
    }
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    return launch_bodies(c, p);
}

// one step of the resident loop: keysReady = the previous step's integrate kernel has written
// this step's cell keys; writeNextKeys = this step's does the same for the next one
static int resident_step(SphCtx *c, const SphParams *p, bool keysReady, bool writeNextKeys)
{
    int rc;
    if (keysReady) {
        if ((rc = sync_params(c, p))) return rc;
        c->haveKeys = true;
    } else if ((rc = sph_hash_position(c, p))) {
        return rc;
    }
    if ((rc = sort_and_table(c))) return rc;
    // the plan needs the tables only: it runs on the second stream beside the reorder pass
    // (one is bound by latency, the other by HBM)
    const bool side = c->stream2 != nullptr;
    if (side) {
        HIPCHK(c, hipEventRecord(c->evTable, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->evTable, 0));
        if ((rc = ensure_plan(c, p, c->stream2))) return rc;
        HIPCHK(c, hipEventRecord(c->evPlan, c->stream2));
    }
    if ((rc = sph_reorder(c))) return rc;
    if (side) HIPCHK(c, hipStreamWaitEvent(c->stream, c->evPlan, 0));
    if ((rc = sph_compute_density_and_pressure(c, p))) return rc;
    if ((rc = sph_compute_force(c, p))) return rc;
    {
        StageTimer t(c, SPH_T_INTEGRATE);
        if ((rc = launch_apply_integrate(c, p, true, true, writeNextKeys))) return rc;
    }
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    if ((rc = launch_bodies(c, p))) return rc;
    c->tm.steps++;
    return SPH_OK;
}

int sph_step_resident(SphCtx *c, const SphParams *p, int32_t nsteps)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->slab && c->anyDynamicBox)
        return fail(c, SPH_E_INVALID, "slab mode with dynamic boxes: use sph_slab_step_particles, "
                                      "sum the impulses over ranks, sph_slab_step_bodies");
    // Inside the loop the integrate kernel also writes the next step's cell keys (it holds the
    // new position in registers), which saves the k_keys pass; not with the parity outputs of
    // SPH_DEBUG_HASHES and not in slab mode (dead ghosts carry their own key).
    const bool fuse = !c->slab && !(c->debug & SPH_DEBUG_HASHES);
    bool keysReady = false;
    for (int s = 0; s < nsteps; s++) {
        if (c->timing) drain_timings(c);
        const bool next = fuse && s + 1 < nsteps;
        if ((rc = resident_step(c, p, keysReady, next))) return rc;
        keysReady = next;
    }
    return SPH_OK;
}

static int read_ecs(SphCtx *c, int32_t buffer, void *out);

int sph_step(SphCtx *c, const SphParams *p, float *pos, float *vel, const float *invMass,
             const SphParticle *params, int32_t N, float *densities_out, float *forces_out)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if ((rc = sph_upload(c, pos, vel, invMass, params, N, p->gridSize))) return rc;
    if ((rc = sph_hash_position(c, p))) return rc;
    if ((rc = sort_and_table(c))) return rc;
    if ((rc = sph_reorder(c))) return rc;
    if ((rc = sph_compute_density_and_pressure(c, p))) return rc;
    if ((rc = sph_compute_force(c, p))) return rc;
    if (densities_out && (rc = read_ecs(c, SPH_BUF_DENSITIES, densities_out))) return rc;
    if (forces_out && (rc = read_ecs(c, SPH_BUF_FORCES, forces_out))) return rc;
    {
        StageTimer t(c, SPH_T_INTEGRATE);
        if ((rc = launch_apply_integrate(c, p, true, true))) return rc;
    }
    if ((rc = launch_bodies(c, p))) return rc;
    c->tm.steps++;
    rc = sph_download(c, pos, vel);
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    return rc;
}

int sph_step_compat(SphCtx *c, const SphParams *p, const float *pos, const float *vel,
                    const float *invMass, const SphParticle *params, int32_t N,
                    float *densities_out, float *forces_out)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (!densities_out || !forces_out) return fail(c, SPH_E_INVALID, "output arrays are NULL");
    if ((rc = sph_upload(c, pos, vel, invMass, params, N, p->gridSize))) return rc;
    if ((rc = sph_hash_position(c, p))) return rc;
    if ((rc = sort_and_table(c))) return rc;
    if ((rc = sph_reorder(c))) return rc;
    if ((rc = sph_compute_density_and_pressure(c, p))) return rc;
    if ((rc = sph_compute_force(c, p))) return rc;
    if ((rc = read_ecs(c, SPH_BUF_DENSITIES, densities_out))) return rc;
    if ((rc = read_ecs(c, SPH_BUF_FORCES, forces_out))) return rc;
    return check_plan_error(c);
}

// ---- inspection ---------------------------------------------------------------------------
static int read_ecs(SphCtx *c, int32_t buffer, void *out)
{
    const int N = c->n, cur = c->cur;
    int rc = ensure(c, c->io3a, (size_t)3 * N);
    if (!rc) rc = ensure(c, c->io1, (size_t)N);
    if (rc) return rc;
    const int nb = blocks_for(N, 256);
    size_t bytes = 0;
    void *src = nullptr;
    switch (buffer) {
    case SPH_BUF_DENSITIES:
        k_unpack1<float><<<nb, 256, 0, c->stream>>>(c->rho.p, c->oid[cur].p, N, c->io1.p);
        src = c->io1.p; bytes = (size_t)4 * N; break;
    case SPH_BUF_PRESSURES:
        k_unpack1<float><<<nb, 256, 0, c->stream>>>(c->prs.p, c->oid[cur].p, N, c->io1.p);
        src = c->io1.p; bytes = (size_t)4 * N; break;
    case SPH_BUF_DENSITY_COUNTS:
        k_unpack1<int><<<nb, 256, 0, c->stream>>>(c->dcount.p, c->oid[cur].p, N, (int *)c->io1.p);
        src = c->io1.p; bytes = (size_t)4 * N; break;
    case SPH_BUF_FORCE_COUNTS:
        k_unpack1<int><<<nb, 256, 0, c->stream>>>(c->fcount.p, c->oid[cur].p, N, (int *)c->io1.p);
        src = c->io1.p; bytes = (size_t)4 * N; break;
    case SPH_BUF_FORCES:
        k_unpack3<<<nb, 256, 0, c->stream>>>(c->force.p, c->oid[cur].p, N, c->io3a.p);
        src = c->io3a.p; bytes = (size_t)12 * N; break;
    case SPH_BUF_POSITIONS:
        k_unpack3<<<nb, 256, 0, c->stream>>>(c->posm[cur].p, c->oid[cur].p, N, c->io3a.p);
        src = c->io3a.p; bytes = (size_t)12 * N; break;
    case SPH_BUF_VELOCITIES:
        k_unpack3<<<nb, 256, 0, c->stream>>>(c->velq[cur].p, c->oid[cur].p, N, c->io3a.p);
        src = c->io3a.p; bytes = (size_t)12 * N; break;
    default: return fail(c, SPH_E_INVALID, "buffer %d is not an ECS-order buffer", buffer);
    }
    HIPCHK(c, hipGetLastError());
    pin_host(c, out, bytes);
    return copy_d2h(c, out, src, bytes);
}

// SPH_BUF_MASK_COUNTS / SPH_BUF_MASK_IDSUMS: the production density pass's hit masks, walked as the
// force pass walks them (k_audit_masks), per particle in ECS order
static int read_mask_audit(SphCtx *c, int32_t buffer, void *out, size_t bytes)
{
    const int N = c->n, cur = c->cur;
    if (!c->havePlan || !c->haveMasks || !c->ordered)
        return fail(c, SPH_E_INVALID, "no hit masks: run the production density pass (no SPH_DEBUG_COUNTS) first");
    const size_t need = (buffer == SPH_BUF_MASK_COUNTS ? 4 : 8) * (size_t)N;
    if (bytes != need) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", need);
    DevBuf<int> cnt;
    DevBuf<unsigned long long> sum, sumEcs;
    DevBuf<uint32_t> inv;
    int rc = ensure(c, cnt, (size_t)N, false);
    if (!rc) rc = ensure(c, sum, (size_t)N, false);
    if (!rc) rc = ensure(c, sumEcs, (size_t)N, false);
    if (!rc) rc = ensure(c, inv, 1, false);
    if (!rc) rc = ensure(c, c->io1, (size_t)N);
    uint32_t invalid = 0;
    if (!rc) {
        SphParams dummy{};
        NbArgs A = neighbor_args(c, &dummy, NB_CAP);
        hipError_t e = hipMemsetAsync(cnt.p, 0, (size_t)4 * N, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(sum.p, 0, (size_t)8 * N, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(inv.p, 0, 4, c->stream);
        if (e == hipSuccess) {
            k_audit_masks<<<2048, 64, 0, c->stream>>>(A, c->oid[cur].p, c->nb > 0 ? c->bperm : nullptr,
                                                      (uint32_t)N, cnt.p, sum.p, inv.p);
            k_audit_gcells<<<1024, 64, 0, c->stream>>>(A, c->oid[cur].p, c->nb > 0 ? c->bperm : nullptr,
                                                       (uint32_t)N, cnt.p, sum.p, inv.p);
            const int nb = blocks_for(N, 256);
            if (buffer == SPH_BUF_MASK_COUNTS) {
                k_unpack1<int><<<nb, 256, 0, c->stream>>>(cnt.p, c->oid[cur].p, N, (int *)c->io1.p);
                e = hipMemcpyAsync(out, c->io1.p, need, hipMemcpyDeviceToHost, c->stream);
            } else {
                k_unpack1<unsigned long long><<<nb, 256, 0, c->stream>>>(sum.p, c->oid[cur].p, N, sumEcs.p);
                e = hipMemcpyAsync(out, sumEcs.p, need, hipMemcpyDeviceToHost, c->stream);
            }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&invalid, inv.p, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = fail(c, SPH_E_DEVICE, "mask audit: %s", hipGetErrorString(e));
    }
    release(cnt); release(sum); release(sumEcs); release(inv);
    if (!rc && invalid)
        rc = fail(c, SPH_E_DEVICE, "mask audit: %u hit bits point outside their block's image", invalid);
    return rc;
}

int sph_read(SphCtx *c, int32_t buffer, void *out, size_t bytes)
{
    if (!c || !out) return SPH_E_INVALID;
    if (c->n <= 0) return fail(c, SPH_E_INVALID, "no particles uploaded");
    if (c->slab) // ids are global there, not indices into the caller's arrays
        return fail(c, SPH_E_INVALID, "slab mode: the ECS-ordered views are not available "
                                      "(sph_slab_download returns the owned particles)");
    HIPCHK(c, hipSetDevice(c->device));
    if (buffer >= SPH_BUF_DENSITIES && buffer <= SPH_BUF_FORCE_COUNTS) // results of the neighbour kernels
        if (int rc_ = check_plan_error(c)) return rc_;
    const size_t N = (size_t)c->n;
    const void *src = nullptr;
    size_t need = 0;
    bool ok = true;
    switch (buffer) {
    case SPH_BUF_HASHES: src = c->hashes.p; need = 4 * N; ok = (c->debug & SPH_DEBUG_HASHES) && c->hashes.p; break;
    case SPH_BUF_CELLS: src = c->cells.p; need = 12 * N; ok = (c->debug & SPH_DEBUG_HASHES) && c->cells.p; break;
    case SPH_BUF_KEYS: src = c->keysEcs.p; need = 4 * N; ok = (c->debug & SPH_DEBUG_HASHES) && c->keysEcs.p; break;
    case SPH_BUF_PERM: src = c->oid[c->cur].p; need = 4 * N; ok = c->ordered; break;
    case SPH_BUF_SORTED_KEYS: src = c->sortedKeys; need = 4 * N; ok = c->haveSort; break;
    case SPH_BUF_TABLE: src = c->table.p; need = (size_t)8 * c->grid.cells; ok = c->haveTable; break;
    case SPH_BUF_BOUNDARY_PERM: src = c->bperm; need = (size_t)4 * c->nb; ok = !c->boundaryDirty && c->nb > 0; break;
    case SPH_BUF_BOUNDARY_TABLE: src = c->btable.p; need = (size_t)8 * c->grid.cells; ok = !c->boundaryDirty; break;
    case SPH_BUF_DENSITIES: case SPH_BUF_PRESSURES:
        if (!c->haveDensity) return fail(c, SPH_E_INVALID, "densities not computed");
        if (bytes != 4 * N) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", 4 * N);
        return read_ecs(c, buffer, out);
    case SPH_BUF_DENSITY_COUNTS:
        if (!c->haveDensity || !(c->debug & SPH_DEBUG_COUNTS)) return fail(c, SPH_E_INVALID, "density counts not computed (SPH_DEBUG_COUNTS)");
        if (bytes != 4 * N) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", 4 * N);
        return read_ecs(c, buffer, out);
    case SPH_BUF_FORCE_COUNTS:
        if (!c->haveForce || !(c->debug & SPH_DEBUG_COUNTS)) return fail(c, SPH_E_INVALID, "force counts not computed (SPH_DEBUG_COUNTS)");
        if (bytes != 4 * N) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", 4 * N);
        return read_ecs(c, buffer, out);
    case SPH_BUF_FORCES:
        if (!c->haveForce) return fail(c, SPH_E_INVALID, "forces not computed");
        if (bytes != 12 * N) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", 12 * N);
        return read_ecs(c, buffer, out);
    case SPH_BUF_POSITIONS: case SPH_BUF_VELOCITIES:
        if (bytes != 12 * N) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", 12 * N);
        return read_ecs(c, buffer, out);
    case SPH_BUF_MASK_COUNTS: case SPH_BUF_MASK_IDSUMS:
        if (int rc_ = check_plan_error(c)) return rc_;
        return read_mask_audit(c, buffer, out, bytes);
    default: return fail(c, SPH_E_INVALID, "unknown buffer %d", buffer);
    }
    if (!ok || !src) return fail(c, SPH_E_INVALID, "buffer %d is not available at this point", buffer);
    if (bytes != need) return fail(c, SPH_E_INVALID, "size mismatch: need %zu bytes", need);
    HIPCHK(c, hipMemcpyAsync(out, src, need, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_sample_density(SphCtx *c, const SphParams *p, const float origin[3], float spacing,
                       const int32_t dims[3], float *field_out)
{
    int rc = sync_params(c, p);
    if (rc) return rc;
    if (!origin || !dims || !field_out || !(spacing > 0.f) || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0)
        return fail(c, SPH_E_INVALID, "sph_sample_density: bad grid");
    const long long nodes = (long long)dims[0] * dims[1] * dims[2];
    if (nodes > (1ll << 31) - 1) return fail(c, SPH_E_TOOLARGE, "sph_sample_density: more than 2^31 nodes");
    if (c->slab) return fail(c, SPH_E_INVALID, "sph_sample_density is not available in slab mode");
    // the table of the resident order (after a step the particles have moved: sort again)
    if (!c->ordered) {
        if ((rc = sph_hash_position(c, p))) return rc;
        if ((rc = sph_radix_sort(c))) return rc;
        if ((rc = sph_map_table(c))) return rc;
        if ((rc = sph_reorder(c))) return rc;
    }
    if ((rc = ensure(c, c->io1, (size_t)std::max<long long>(nodes, c->n)))) return rc;
    const int cur = c->cur;
    k_sample_density<<<blocks_for((long)nodes, 256), 256, 0, c->stream>>>(
        c->posm[cur].p, c->mat[cur].p, c->dmats.p, c->table.p, c->grid, origin[0], origin[1], origin[2],
        spacing, dims[0], dims[1], dims[2], c->io1.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(field_out, c->io1.p, sizeof(float) * (size_t)nodes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_find_bounds(SphCtx *c, float gridSize, float mn[3], float mx[3])
{
    if (!c || !mn || !mx) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (c->n <= 0) return fail(c, SPH_E_INVALID, "no particles uploaded");
    int rc = bounds_of(c, c->posm[c->cur].p, c->n, mn, mx);
    if (rc) return rc;
    if (c->nb > 0) {
        float bmn[3], bmx[3];
        rc = bounds_of(c, c->bposmIn.p, c->nb, bmn, bmx);
        if (rc) return rc;
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], bmn[a]);
            mx[a] = std::max(mx[a], bmx[a]);
        }
    }
    for (int a = 0; a < 3; a++) { // FluidSPHUtils.cs:85-92
        volatile float lo = mn[a] / gridSize, hi = mx[a] / gridSize;
        volatile float l = floorf(lo) * gridSize, h = ceilf(hi) * gridSize;
        mn[a] = l;
        mx[a] = h + 1.0f;
    }
    return SPH_OK;
}

int sph_sort_pairs(SphCtx *c, const uint32_t *keys, int32_t n, int32_t bits, uint32_t *perm_out,
                   uint32_t *keys_out)
{
    if (!c) return SPH_E_INVALID;
    if (n < 0 || bits < 1 || bits > 32 || (n > 0 && !keys))
        return fail(c, SPH_E_INVALID, "bad sort arguments");
    if (n == 0) return SPH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf<uint32_t> ka, kb, va, vb;
    int rc = ensure(c, ka, (size_t)n, false);
    if (!rc) rc = ensure(c, kb, (size_t)n, false);
    if (!rc) rc = ensure(c, va, (size_t)n, false);
    if (!rc) rc = ensure(c, vb, (size_t)n, false);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(ka.p, keys, (size_t)4 * n, hipMemcpyHostToDevice, c->stream);
        uint32_t *sk = nullptr, *sv = nullptr;
        if (e == hipSuccess) rc = radix_sort(c, ka.p, va.p, kb.p, vb.p, n, bits, &sk, &sv);
        if (e == hipSuccess && !rc && perm_out)
            e = hipMemcpyAsync(perm_out, sv, (size_t)4 * n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && !rc && keys_out)
            e = hipMemcpyAsync(keys_out, sk, (size_t)4 * n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && !rc) rc = fail(c, SPH_E_DEVICE, "sort: %s", hipGetErrorString(e));
    }
    release(ka); release(kb); release(va); release(vb);
    return rc;
}

// Stable LSD radix sort of device keys into (keys_out, perm_out) with every intermediate in the
// static scratch: pass p writes to the caller's outputs or to the scratch pair, alternating so
// that the last pass lands in the outputs.  No hipMalloc / hipFree here or below.
int sph_sort_pairs_noalloc(SphCtx *c, const uint32_t *keys_dev, int32_t n, int32_t bits,
                           uint32_t *perm_out_dev, uint32_t *keys_out_dev)
{
    if (!c) return SPH_E_INVALID;
    if (n < 0 || bits < 1 || bits > 32 || (n > 0 && (!keys_dev || !perm_out_dev || !keys_out_dev)))
        return fail(c, SPH_E_INVALID, "bad sort arguments");
    if (n > SPH_SORT_NOALLOC_MAX)
        return fail(c, SPH_E_TOOLARGE, "sph_sort_pairs_noalloc: n = %d beyond the static scratch "
                                       "(at most %d keys)", n, SPH_SORT_NOALLOC_MAX);
    if (n == 0) return SPH_OK;
    if (keys_dev == keys_out_dev || (const void *)keys_dev == (const void *)perm_out_dev ||
        keys_out_dev == perm_out_dev)
        return fail(c, SPH_E_INVALID, "sph_sort_pairs_noalloc: buffers must not alias");
    if (c->device >= SPH_MAX_DEVICES || !g_scratchPtr[c->device])
        return fail(c, SPH_E_DEVICE, "no static scratch on device %d", c->device);
    HIPCHK(c, hipSetDevice(c->device));
    SortScratch *sc = g_scratchPtr[c->device]; // device address, never the symbol's name
    const int nblocks = blocks_for(n, RS_TILE);
    const int passes = (bits + 7) / 8;
    uint32_t *dk[2] = {keys_out_dev, sc->keys}, *dv[2] = {perm_out_dev, sc->vals};
    const uint32_t *kin = keys_dev, *vin = nullptr; // values of pass 0 are the indices
    for (int p = 0; p < passes; p++) {
        const int d = (p + passes - 1) & 1; // the last pass has d == 0: the caller's outputs
        // hist and rowTotal are written in full by k_radix_hist / k_radix_scan before they are
        // read, so nothing of the previous call's contents has to be cleared
        k_radix_hist<<<nblocks, RS_THREADS, 0, c->stream>>>(kin, n, 8 * p, sc->hist, nblocks);
        k_radix_scan<<<256, RS_THREADS, 0, c->stream>>>(sc->hist, sc->rowTotal, nblocks);
        k_radix_scatter<<<nblocks, RS_THREADS, 0, c->stream>>>(kin, vin, dk[d], dv[d], n, 8 * p,
                                                                sc->hist, sc->rowTotal, nblocks);
        kin = dk[d];
        vin = dv[d];
    }
    HIPCHK(c, hipGetLastError());
    return SPH_OK;
}

int sph_get_timings(SphCtx *c, SphTimings *out)
{
    if (!c || !out) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    int rc = collect_timings(c);
    if (rc) return rc;
    *out = c->tm;
    return SPH_OK;
}

int sph_reset_timings(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    int rc = collect_timings(c);
    if (rc) return rc;
    std::memset(&c->tm, 0, sizeof(c->tm));
    return SPH_OK;
}

int sph_seed_volume(SphCtx *c, const float localToWorld[12], const float boundsMin[3],
                    const float boundsSize[3], float radius, const SphParticle *params,
                    float invMass, float gridSize, int32_t append, int32_t *added)
{
    if (!c || !localToWorld || !boundsMin || !boundsSize || !params || !added) return SPH_E_INVALID;
    if (int rc_ = use_device(c)) return rc_;
    if (c->slab) return fail(c, SPH_E_INVALID, "sph_seed_volume is not available in slab mode");
    if (!(radius > 0.f) || !(invMass > 0.f) || !(gridSize > 0.f) || !std::isfinite(gridSize))
        return fail(c, SPH_E_INVALID, "radius, invMass and gridSize must be positive");
    c->matAllZero = false; // (seeded particles carry their own material index)
    SeedDesc d;
    long long count = 1;
    int num[3];
    for (int a = 0; a < 3; a++) {
        // radius / |localToWorld * axis|  (:65-67), (int)((size + r) / (2 r))  (:69-71)
        const float x = localToWorld[a], y = localToWorld[4 + a], z = localToWorld[8 + a];
        volatile float len2 = (x * x + y * y) + z * z;
        const float len = sqrtf(len2);
        d.rad[a] = radius / len;
        d.bmin[a] = boundsMin[a];
        if (!(d.rad[a] > 0.f) || !std::isfinite(d.rad[a]))
            return fail(c, SPH_E_INVALID, "degenerate localToWorld");
        const float q = (boundsSize[a] + d.rad[a]) / (2.0f * d.rad[a]);
        num[a] = q > 0.f ? (int)std::min(q, 1.0e6f) : 0;
        count *= num[a];
    }
    std::memcpy(d.l2w, localToWorld, sizeof(d.l2w));
    d.nx = num[0]; d.ny = num[1]; d.nz = num[2];
    *added = 0;
    const int base = append ? c->n : 0;
    if (count <= 0) return SPH_OK;
    if (base + count > (1ll << 30)) return fail(c, SPH_E_TOOLARGE, "more than 2^30 particles");
    HIPCHK(c, hipSetDevice(c->device));
    const int cur = append ? c->cur : 0;
    const size_t total = (size_t)(base + count);
    int rc = grow_keep(c, c->posm[cur], total, (size_t)base);
    if (!rc) rc = grow_keep(c, c->velq[cur], total, (size_t)base);
    if (!rc) rc = grow_keep(c, c->mat[cur], total, (size_t)base);
    if (!rc) rc = grow_keep(c, c->oid[cur], total, (size_t)base);
    if (!rc) rc = ensure_fluid(c, (int)total);
    if (rc) return rc;
    // material of the new particles
    if (!append) {
        c->matKeys.clear();
        c->matIndex.clear();
    }
    MatKey k;
    std::memcpy(&k.w[0], params, 16);
    std::memcpy(&k.w[4], &invMass, 4);
    uint32_t mi = 0;
    for (; mi < c->matKeys.size(); mi++)
        if (c->matKeys[mi] == k) break;
    if (mi == c->matKeys.size()) c->matKeys.push_back(k);
    rc = upload_mats(c, c->matRate > 0.f ? c->matRate : 4.0f);
    if (rc) return rc;
    k_seed_lattice<<<blocks_for((long)count, 256), 256, 0, c->stream>>>(
        d, c->mats[mi].mass, mi, (uint32_t)base, (int)count, c->posm[cur].p + base,
        c->velq[cur].p + base, c->mat[cur].p + base, c->oid[cur].p + base);
    HIPCHK(c, hipGetLastError());
    c->n = (int)total;
    c->cur = cur;
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    c->rhoInOrder = false;
    if (c->gridExplicit) {
        const int o[3] = {c->grid.ox, c->grid.oy, c->grid.oz};
        const int dd[3] = {c->grid.gx, c->grid.gy, c->grid.gz};
        set_grid(c, o, dd, gridSize);
    } else {
        rc = auto_grid(c, gridSize);
        if (rc) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *added = (int32_t)count;
    return ensure_debug(c);
}

// ---- slab decomposition across GPUs ------------------------------------------
int sph_slab_upload(SphCtx *c, const float *pos, const float *vel, const uint32_t *materialIndex,
                    const uint32_t *ids, int32_t N, const float *materials, int32_t M,
                    float gridSize, int32_t capacity)
{
    if (!c) return SPH_E_INVALID;
    if (N < 0 || (N > 0 && (!pos || !vel || !materialIndex || !ids)) || !materials || M <= 0)
        return fail(c, SPH_E_INVALID, "sph_slab_upload: bad arrays");
    c->matAllZero = false;
    if (capacity < std::max(N, 1)) return fail(c, SPH_E_INVALID, "capacity below N");
    if (!(gridSize > 0.f) || !std::isfinite(gridSize))
        return fail(c, SPH_E_INVALID, "gridSize must be positive and finite");
    for (int i = 0; i < N; i++) {
        if (materialIndex[i] >= (uint32_t)M) return fail(c, SPH_E_INVALID, "material index out of range");
        if (ids[i] > SPH_ID_MASK) return fail(c, SPH_E_TOOLARGE, "particle ids are limited to 30 bits");
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_fluid(c, capacity);
    if (!rc) rc = ensure(c, c->io3a, (size_t)3 * std::max(N, 1));
    if (!rc) rc = ensure(c, c->io3b, (size_t)3 * std::max(N, 1));
    if (!rc) rc = ensure(c, c->ioU, (size_t)2 * std::max(N, 1));
    if (!rc) rc = ensure(c, c->slabCounters, 4, false);
    if (!rc) rc = ensure(c, c->slabNext, 4, false);
    if (rc) return rc;
    // the material table is the caller's, in the caller's order (the same on every rank)
    c->matKeys.clear();
    c->matIndex.clear();
    for (int m = 0; m < M; m++) {
        MatKey k;
        std::memcpy(k.w, materials + 5 * (size_t)m, 20); // radius, rho0, mu, k, invMass
        c->matKeys.push_back(k);
    }
    rc = upload_mats(c, c->matRate > 0.f ? c->matRate : 4.0f);
    if (rc) return rc;
    c->slab = true;
    c->slabKeys = false;
    c->nextCounts = false;
    c->expectHalo = -1.f;
    c->slabCap = capacity;
    c->n = N;
    c->nOwned = N;
    c->nGhost = c->nDead = 0;
    c->cur = 0;
    if (N > 0) {
        HIPCHK(c, hipMemcpyAsync(c->io3a.p, pos, (size_t)12 * N, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->io3b.p, vel, (size_t)12 * N, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->ioU.p, materialIndex, (size_t)4 * N, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->ioU.p + N, ids, (size_t)4 * N, hipMemcpyHostToDevice, c->stream));
        k_slab_pack_fluid<<<blocks_for(N, 256), 256, 0, c->stream>>>(
            c->io3a.p, c->io3b.p, c->ioU.p, c->ioU.p + N, c->dmats.p, N, c->posm[0].p,
            c->velq[0].p, c->mat[0].p, c->oid[0].p);
        HIPCHK(c, hipGetLastError());
    }
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->gridExplicit) {
        // the table must be the same on every rank only in cell size; its extent is local
        if (N > 0) {
            rc = auto_grid(c, gridSize);
            if (rc) return rc;
        } else {
            const int o[3] = {0, 0, 0}, d[3] = {64, 64, 64};
            set_grid(c, o, d, gridSize);
        }
    } else {
        const int o[3] = {c->grid.ox, c->grid.oy, c->grid.oz};
        const int d[3] = {c->grid.gx, c->grid.gy, c->grid.gz};
        set_grid(c, o, d, gridSize);
    }
    return SPH_OK;
}

// The counters of an exchange with these cuts: counted by the integrate kernel of the step before
// when the cuts are the ones of the last exchange (fixed cuts), else by a pass of their own
// (k_slab_pack<true>: nothing is written, no record changes state).  Leaves them in
// c->slabCounters; remembers the cuts for the next integrate.
static int slab_count(SphCtx *c, float xlo, float xhi, float halo)
{
    const int cur = c->cur;
    const bool ready = c->nextCounts && !c->slabCountPass && xlo == c->expectLo && xhi == c->expectHi &&
                       halo == c->expectHalo;
    c->nextCounts = false;
    c->expectLo = xlo;
    c->expectHi = xhi;
    c->expectHalo = halo;
    if (ready) {
        HIPCHK(c, hipMemcpyAsync(c->slabCounters.p, c->slabNext.p, 16, hipMemcpyDeviceToDevice, c->stream));
        return SPH_OK;
    }
    HIPCHK(c, hipMemsetAsync(c->slabCounters.p, 0, 16, c->stream));
    if (c->n > 0)
        k_slab_pack<true><<<blocks_for(c->n, 256), 256, 0, c->stream>>>(
            c->posm[cur].p, c->velq[cur].p, c->mat[cur].p, c->oid[cur].p, c->n, xlo, xhi, halo,
            nullptr, nullptr, 0, c->slabCounters.p, nullptr, c->grid);
    HIPCHK(c, hipGetLastError());
    return SPH_OK;
}

int sph_slab_pack(SphCtx *c, float xlo, float xhi, float halo, void *left_dev, void *right_dev,
                  int32_t capRecords, int32_t counts[4])
{
    if (!c || !counts) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_INVALID, "sph_slab_pack before sph_slab_upload");
    if (capRecords < 0 || (capRecords > 0 && (!left_dev || !right_dev)))
        return fail(c, SPH_E_INVALID, "sph_slab_pack: bad buffers");
    if (!(halo >= 0.f) || !(xlo <= xhi)) return fail(c, SPH_E_INVALID, "sph_slab_pack: bad slab");
    HIPCHK(c, hipSetDevice(c->device));
    // sizing pass first: an exchange buffer that is too small is reported with the counts it
    // would need and with nothing packed, flagged or re-counted, so the caller can grow it and
    // call again
    const int cur = c->cur;
    uint32_t h[4] = {0, 0, 0, 0};
    if (c->n > 0) {
        if (int rc = slab_count(c, xlo, xhi, halo)) return rc;
        HIPCHK(c, hipMemcpyAsync(h, c->slabCounters.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 4; i++) counts[i] = (int32_t)h[i];
        if (h[0] > (uint32_t)capRecords || h[1] > (uint32_t)capRecords) {
            // (the counts stay good for the caller's second attempt with larger buffers)
            c->nextCounts = true;
            HIPCHK(c, hipMemcpyAsync(c->slabNext.p, c->slabCounters.p, 16, hipMemcpyDeviceToDevice, c->stream));
            return fail(c, SPH_E_TOOLARGE, "exchange buffer too small: %u / %u records, "
                        "capacity %d (nothing was packed; counts[] holds what is needed)",
                        h[0], h[1], capRecords);
        }
        HIPCHK(c, hipMemsetAsync(c->slabCounters.p, 0, 16, c->stream));
        k_slab_pack<false><<<blocks_for(c->n, 256), 256, 0, c->stream>>>(
            c->posm[cur].p, c->velq[cur].p, c->mat[cur].p, c->oid[cur].p, c->n, xlo, xhi, halo,
            (float4 *)left_dev, (float4 *)right_dev, capRecords, c->slabCounters.p, c->keysA.p, c->grid);
        HIPCHK(c, hipGetLastError());
    }
    if (c->n <= 0)
        for (int i = 0; i < 4; i++) counts[i] = 0;
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the caller sends the buffers from its own stream
    // last step's ghosts are dead now; the particles that left stay as this step's ghosts
    const int moved = (int)(h[2] + h[3]);
    c->nDead += c->nGhost;
    c->nGhost = moved;
    c->nOwned -= moved;
    c->slabKeys = true; // (the pack pass has written the keys of every resident record)
    return SPH_OK;
}

int sph_slab_import(SphCtx *c, const void *records_dev, int32_t count, int32_t transfers)
{
    if (!c) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_INVALID, "sph_slab_import before sph_slab_upload");
    if (count < 0 || transfers < 0 || transfers > count || (count > 0 && !records_dev))
        return fail(c, SPH_E_INVALID, "sph_slab_import: bad arguments");
    if (count == 0) return SPH_OK;
    if ((long long)c->n + count > (1ll << 30))
        return fail(c, SPH_E_TOOLARGE, "slab beyond 2^30 records (%d + %d)", c->n, count);
    HIPCHK(c, hipSetDevice(c->device));
    const int cur = c->cur, n = c->n;
    if (n + count > c->slabCap) { // grow, keeping the resident records
        const size_t want = (size_t)(n + count) + (size_t)(n + count) / 4 + 1024;
        int rc = grow_keep(c, c->posm[cur], want, (size_t)n);
        if (!rc) rc = grow_keep(c, c->velq[cur], want, (size_t)n);
        if (!rc) rc = grow_keep(c, c->mat[cur], want, (size_t)n);
        if (!rc) rc = grow_keep(c, c->oid[cur], want, (size_t)n);
        if (!rc) rc = ensure_fluid(c, (int)want);
        if (rc) return rc;
        c->slabCap = (int)want;
        c->slabKeys = false; // (the key buffer has been reallocated: k_slab_keys for this step)
    }
    k_slab_import<<<blocks_for(count, 256), 256, 0, c->stream>>>(
        (const float4 *)records_dev, count, c->dmats.p, (int)c->mats.size(), c->posm[cur].p + n,
        c->velq[cur].p + n, c->mat[cur].p + n, c->oid[cur].p + n, c->keysA.p + n, c->grid);
    HIPCHK(c, hipGetLastError());
    c->rhoInOrder = false;
    c->nextCounts = false; // (records the integrate kernel has not seen)
    c->n += count;
    c->nOwned += transfers;
    c->nGhost += count - transfers;
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    return SPH_OK;
}

int sph_slab_step_particles(SphCtx *c, const SphParams *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (!c->slab) return fail(c, SPH_E_INVALID, "sph_slab_step_particles before sph_slab_upload");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = sph_hash_position(c, p))) return rc;
    if ((rc = sort_and_table(c))) return rc;
    if ((rc = sph_reorder(c))) return rc;
    if ((rc = sph_compute_density_and_pressure(c, p))) return rc;
    if ((rc = sph_compute_force(c, p))) return rc;
    {
        StageTimer t(c, SPH_T_INTEGRATE);
        if ((rc = launch_apply_integrate(c, p, true, true))) return rc;
    }
    c->haveKeys = c->haveSort = c->haveTable = c->ordered = false;
    c->haveDensity = c->haveForce = false;
    c->tm.steps++;
    return SPH_OK;
}

int sph_slab_get_impulses(SphCtx *c, float *impulses_out, int32_t K)
{
    if (!c || (K > 0 && !impulses_out)) return SPH_E_INVALID;
    if (K != c->nboxes) return fail(c, SPH_E_INVALID, "K != number of colliders");
    if (K == 0) return SPH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(impulses_out, c->bodyImpulse.p, sizeof(float) * 6 * (size_t)K,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_slab_step_bodies(SphCtx *c, const SphParams *p, const float *impulses, int32_t K)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (K != c->nboxes) return fail(c, SPH_E_INVALID, "K != number of colliders");
    if (K == 0) return SPH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (impulses)
        HIPCHK(c, hipMemcpyAsync(c->bodyImpulse.p, impulses, sizeof(float) * 6 * (size_t)K,
                                 hipMemcpyHostToDevice, c->stream));
    rc = launch_bodies(c, p);
    if (rc) return rc;
    if (impulses) HIPCHK(c, hipStreamSynchronize(c->stream)); // the host array may go away
    return SPH_OK;
}

// ---- native exchange over RCCL (SURVEY.md section 5 "distributed communication backend", 8e) ----
// librccl is opened on first use, so that a process which never exchanges does not load it.
namespace {
struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::mutex g_rcclLock;

const char *rccl_open()
{
    std::lock_guard<std::mutex> lk(g_rcclLock);
    if (g_rccl.ok) return nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        g_rccl.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.so) break;
    }
    if (!g_rccl.so) return "librccl.so not found";
#define RCCL_SYM(field, sym)                                                                     \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.so, #sym);                               \
    if (!g_rccl.field) return "librccl.so lacks " #sym;
    RCCL_SYM(GetUniqueId, ncclGetUniqueId)
    RCCL_SYM(CommInitRank, ncclCommInitRank)
    RCCL_SYM(CommDestroy, ncclCommDestroy)
    RCCL_SYM(GroupStart, ncclGroupStart)
    RCCL_SYM(GroupEnd, ncclGroupEnd)
    RCCL_SYM(Send, ncclSend)
    RCCL_SYM(Recv, ncclRecv)
    RCCL_SYM(CommGetAsyncError, ncclCommGetAsyncError)
    RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef RCCL_SYM
    g_rccl.ok = true;
    return nullptr;
}
} // namespace

#define RCCLCHK(c, expr)                                                                         \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess)                                                                   \
            return fail(c, SPH_E_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(r_));            \
    } while (0)
// the same inside ncclGroupStart .. ncclGroupEnd: the group is closed before the error is returned
#define RCCLCHK_IN_GROUP(c, expr)                                                                \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            (void)g_rccl.GroupEnd();                                                             \
            return fail(c, SPH_E_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(r_));            \
        }                                                                                        \
    } while (0)

int sph_slab_comm_unique_id(uint8_t *id_out, int32_t cap)
{
    if (!id_out || cap < (int32_t)sizeof(ncclUniqueId)) return SPH_E_INVALID;
    if (const char *e = rccl_open()) return fail(nullptr, SPH_E_DEVICE, "%s", e);
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return fail(nullptr, SPH_E_DEVICE, "ncclGetUniqueId failed");
    std::memcpy(id_out, &id, sizeof(id));
    return (int)sizeof(id);
}

int sph_slab_comm_init(SphCtx *c, int32_t rank, int32_t world, const uint8_t *unique_id, int32_t id_bytes)
{
    if (!c) return SPH_E_INVALID;
    if (world < 1 || rank < 0 || rank >= world || !unique_id || id_bytes != (int32_t)sizeof(ncclUniqueId))
        return fail(c, SPH_E_INVALID, "sph_slab_comm_init: bad rank / world / unique id");
    if (c->comm) return fail(c, SPH_E_INVALID, "sph_slab_comm_init: communicator already open");
    if (const char *e = rccl_open()) return fail(c, SPH_E_DEVICE, "%s", e);
    if (int rc_ = use_device(c)) return rc_;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    RCCLCHK(c, g_rccl.CommInitRank(&c->comm, world, id, rank));
    c->commRank = rank;
    c->commWorld = world;
    int rc = ensure(c, c->xhdr, 8, false);
    if (rc) return rc;
    if (!c->xhost) HIPCHK(c, hipHostMalloc((void **)&c->xhost, 12 * sizeof(uint32_t), hipHostMallocDefault));
    return SPH_OK;
}

int sph_slab_comm_destroy(SphCtx *c)
{
    if (!c) return SPH_E_INVALID;
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    for (int s = 0; s < 2; s++) { release(c->xsend[s]); release(c->xrecv[s]); }
    release(c->xhdr);
    if (c->xhost) (void)hipHostFree(c->xhost);
    c->xhost = nullptr;
    return SPH_OK;
}

// One step's exchange with the two x neighbours, inside the library: count, trade the counts
// (16 bytes a side), ONE host synchronisation to size the messages, pack, trade the records
// (ncclSend / ncclRecv in one group, device buffers end to end over xGMI), import.  Everything is
// ordered on the context's stream; the import kernels are queued behind the receives and nothing
// waits for them.  counts = records sent left, sent right, received from left, received from right.
int sph_slab_exchange(SphCtx *c, float xlo, float xhi, float halo, int32_t counts[4])
{
    if (!c) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_INVALID, "sph_slab_exchange before sph_slab_upload");
    if (!c->comm) return fail(c, SPH_E_INVALID, "sph_slab_exchange before sph_slab_comm_init");
    if (!(halo >= 0.f) || !(xlo <= xhi)) return fail(c, SPH_E_INVALID, "sph_slab_exchange: bad slab");
    if (int rc_ = use_device(c)) return rc_;
    const int cur = c->cur, left = c->commRank - 1, right = c->commRank + 1;
    const bool hasL = left >= 0, hasR = right < c->commWorld;
    // 1. what this rank will send (no record changes state yet)
    HIPCHK(c, hipMemsetAsync(c->xhdr.p, 0, 32, c->stream));
    if (int rc0 = slab_count(c, xlo, xhi, halo)) return rc0;
    // 2. the counters go to both neighbours as they are: the left one reads words 0 and 2 (records,
    // transfers towards it), the right one words 1 and 3
    if (hasL || hasR) {
        RCCLCHK(c, g_rccl.GroupStart());
        if (hasL) {
            RCCLCHK_IN_GROUP(c, g_rccl.Send(c->slabCounters.p, 16, ncclUint8, left, c->comm, c->stream));
            RCCLCHK_IN_GROUP(c, g_rccl.Recv(c->xhdr.p, 16, ncclUint8, left, c->comm, c->stream));
        }
        if (hasR) {
            RCCLCHK_IN_GROUP(c, g_rccl.Send(c->slabCounters.p, 16, ncclUint8, right, c->comm, c->stream));
            RCCLCHK_IN_GROUP(c, g_rccl.Recv(c->xhdr.p + 4, 16, ncclUint8, right, c->comm, c->stream));
        }
        RCCLCHK(c, g_rccl.GroupEnd());
    }
    HIPCHK(c, hipMemcpyAsync(c->xhost, c->slabCounters.p, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->xhost + 4, c->xhdr.p, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the one host synchronisation of the step
    {
        ncclResult_t async = ncclSuccess;
        RCCLCHK(c, g_rccl.CommGetAsyncError(c->comm, &async));
        if (async != ncclSuccess)
            return fail(c, SPH_E_DEVICE, "RCCL asynchronous error: %s", g_rccl.GetErrorString(async));
    }
    const uint32_t *h = c->xhost;
    // a neighbour sends to its right (words 1, 3) what arrives here from the left, and vice versa
    const uint32_t sendN[2] = {hasL ? h[0] : 0u, hasR ? h[1] : 0u};
    const uint32_t recvN[2] = {hasL ? h[4 + 1] : 0u, hasR ? h[8 + 0] : 0u};
    const uint32_t recvT[2] = {hasL ? h[4 + 3] : 0u, hasR ? h[8 + 2] : 0u};
    // Room for the messages and for what arrives.  Everything that can fail locally is done HERE,
    // before this rank commits to the trade of the records: sph_slab_exchange_reserve sizes the
    // buffers up front so that a step normally allocates nothing; beyond the reserve they grow
    // (a hipFree / hipMalloc in the middle of the run).  A rank that fails here returns while its
    // neighbours wait in ncclRecv -- an out-of-memory condition ends the run on every rank, the
    // launcher's job; it cannot be reported to the peers without a second header trade per step.
    const uint32_t capSend = std::max(std::max(h[0], h[1]), 1u);
    int rc = SPH_OK;
    for (int s = 0; s < 2 && !rc; s++) {
        rc = ensure(c, c->xsend[s], 2 * (size_t)capSend);
        if (!rc) rc = ensure(c, c->xrecv[s], 2 * (size_t)std::max(recvN[s], 1u));
    }
    if (!rc && (long long)c->n + recvN[0] + recvN[1] > (1ll << 30))
        rc = fail(c, SPH_E_TOOLARGE, "slab beyond 2^30 records");
    if (!rc && c->n + (int)(recvN[0] + recvN[1]) > c->slabCap) { // (as sph_slab_import would, later)
        const int cur0 = c->cur, n0 = c->n;
        const size_t want = (size_t)n0 + recvN[0] + recvN[1] + ((size_t)n0 + recvN[0] + recvN[1]) / 4 + 1024;
        rc = grow_keep(c, c->posm[cur0], want, (size_t)n0);
        if (!rc) rc = grow_keep(c, c->velq[cur0], want, (size_t)n0);
        if (!rc) rc = grow_keep(c, c->mat[cur0], want, (size_t)n0);
        if (!rc) rc = grow_keep(c, c->oid[cur0], want, (size_t)n0);
        if (!rc) rc = ensure_fluid(c, (int)want);
        if (!rc) c->slabCap = (int)want;
    }
    if (rc) return rc;
    const uint32_t cap = (uint32_t)std::min<size_t>(std::min(c->xsend[0].cap, c->xsend[1].cap) / 2, 1u << 30);
    // 3. pack: last step's ghosts die, the particles that left stay as this step's ghosts
    HIPCHK(c, hipMemsetAsync(c->slabCounters.p, 0, 16, c->stream));
    if (c->n > 0)
        k_slab_pack<false><<<blocks_for(c->n, 256), 256, 0, c->stream>>>(
            c->posm[cur].p, c->velq[cur].p, c->mat[cur].p, c->oid[cur].p, c->n, xlo, xhi, halo,
            c->xsend[0].p, c->xsend[1].p, (int)cap, c->slabCounters.p, c->keysA.p, c->grid);
    HIPCHK(c, hipGetLastError());
    c->slabKeys = true; // (and the imports below add the keys of what arrives)
    const int moved = (int)((hasL ? h[2] : 0u) + (hasR ? h[3] : 0u));
    // (particles beyond an outer rank's open side are not sent anywhere: xlo / xhi are infinite there)
    c->nDead += c->nGhost;
    c->nGhost = (int)(h[2] + h[3]);
    c->nOwned -= (int)(h[2] + h[3]);
    (void)moved;
    // 4. the records
    if (sendN[0] + sendN[1] + recvN[0] + recvN[1] > 0) {
        RCCLCHK(c, g_rccl.GroupStart());
        if (sendN[0]) RCCLCHK_IN_GROUP(c, g_rccl.Send(c->xsend[0].p, (size_t)sendN[0] * 32, ncclUint8, left, c->comm, c->stream));
        if (sendN[1]) RCCLCHK_IN_GROUP(c, g_rccl.Send(c->xsend[1].p, (size_t)sendN[1] * 32, ncclUint8, right, c->comm, c->stream));
        if (recvN[0]) RCCLCHK_IN_GROUP(c, g_rccl.Recv(c->xrecv[0].p, (size_t)recvN[0] * 32, ncclUint8, left, c->comm, c->stream));
        if (recvN[1]) RCCLCHK_IN_GROUP(c, g_rccl.Recv(c->xrecv[1].p, (size_t)recvN[1] * 32, ncclUint8, right, c->comm, c->stream));
        RCCLCHK(c, g_rccl.GroupEnd());
    }
    // 5. append what arrived (queued behind the receives on the same stream)
    for (int s = 0; s < 2; s++)
        if (recvN[s] && (rc = sph_slab_import(c, c->xrecv[s].p, (int32_t)recvN[s], (int32_t)recvT[s]))) return rc;
    if (counts) {
        counts[0] = (int32_t)sendN[0];
        counts[1] = (int32_t)sendN[1];
        counts[2] = (int32_t)recvN[0];
        counts[3] = (int32_t)recvN[1];
    }
    return SPH_OK;
}

// Sizes the exchange buffers of sph_slab_exchange for `records` records a side (32 bytes each, four
// buffers), so that the steps that follow allocate nothing as long as no message is larger.
int sph_slab_exchange_reserve(SphCtx *c, int32_t records)
{
    if (!c) return SPH_E_INVALID;
    if (records < 0) return fail(c, SPH_E_INVALID, "sph_slab_exchange_reserve: negative size");
    if (int rc_ = use_device(c)) return rc_;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < 2; s++) {
        int rc = ensure(c, c->xsend[s], 2 * (size_t)std::max(records, 1), false);
        if (!rc) rc = ensure(c, c->xrecv[s], 2 * (size_t)std::max(records, 1), false);
        if (rc) return rc;
    }
    return SPH_OK;
}

int sph_slab_counts(SphCtx *c, int32_t counts[4])
{
    if (!c || !counts) return SPH_E_INVALID;
    counts[0] = c->n;
    counts[1] = c->nOwned;
    counts[2] = c->nGhost;
    counts[3] = c->nDead;
    return SPH_OK;
}

int sph_slab_download(SphCtx *c, float *pos_out, float *vel_out, float *density_out,
                      uint32_t *ids_out, int32_t cap, int32_t *count)
{
    if (!c || !count) return SPH_E_INVALID;
    if (!c->slab) return fail(c, SPH_E_INVALID, "sph_slab_download before sph_slab_upload");
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc_ = check_plan_error(c)) return rc_;
    const int n = c->n, cur = c->cur;
    std::vector<float4> hp((size_t)n), hv((size_t)n);
    std::vector<uint32_t> hid((size_t)n);
    std::vector<float> hr((size_t)n);
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(hp.data(), c->posm[cur].p, (size_t)16 * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hv.data(), c->velq[cur].p, (size_t)16 * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hid.data(), c->oid[cur].p, (size_t)4 * n, hipMemcpyDeviceToHost, c->stream));
        if (density_out && c->rhoInOrder)
            HIPCHK(c, hipMemcpyAsync(hr.data(), c->rho.p, (size_t)4 * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    int k = 0;
    for (int i = 0; i < n; i++) {
        if (hid[i] & (SPH_ID_GHOST | SPH_ID_DEAD)) continue;
        if (k < cap) {
            if (pos_out) { pos_out[3 * k] = hp[i].x; pos_out[3 * k + 1] = hp[i].y; pos_out[3 * k + 2] = hp[i].z; }
            if (vel_out) { vel_out[3 * k] = hv[i].x; vel_out[3 * k + 1] = hv[i].y; vel_out[3 * k + 2] = hv[i].z; }
            if (density_out) density_out[k] = c->rhoInOrder ? hr[i] : 0.f;
            if (ids_out) ids_out[k] = hid[i] & SPH_ID_MASK;
        }
        k++;
    }
    *count = k;
    if (k > cap) return fail(c, SPH_E_TOOLARGE, "%d owned particles, room for %d", k, cap);
    return SPH_OK;
}

} // extern "C"
