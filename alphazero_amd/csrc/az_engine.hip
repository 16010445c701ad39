// az_engine.hip -- PUCT self-play engine for gfx950 (MI355X): node pools, search kernels, C ABI.
//
// Every game owns its root position, its node pool and its path stack.  Nothing is shared
// between games, so there are no atomics or inter-workgroup hand-offs on the search path;
// parallelism is across games only and each tree sees its simulations strictly in order,
// which is what makes visit counts equal to the reference's (alphazero/mcts.py:110-158).
// The per-simulation kernels give each tree a group of 8 (Connect Four) or 16 (tic-tac-toe)
// lanes, 8 or 4 trees per wavefront; the once-per-move kernels (root expansion + noise, move
// choice) give each tree a whole wavefront.
//
// Node = (A+1) records of 16 B: edge i = {N u32, W f32, P f32, child u32},
// last record = {num_edges, sum N, -, -}.  For Connect Four that is one
// 128-byte line: one HBM/L2 line per node visited during select.  Lane i loads
// record i with a single dwordx4; PUCT scores are formed per lane in fp64 and
// arg-maxed across the 8/16-lane row.
//
// Numerics are the reference's numpy-2 promotion (alphazero/mcts.py:43-59):
//   Q = (float)((double)W/(double)N); U = (Cs*(double)P)*(sqrt(Ns)/(double)(1+N));
//   score = (double)Q + U.  Cs and sqrt(Ns) come from a host table indexed by Ns.
// The file is compiled with -ffp-contract=off: a fused multiply-add anywhere in
// the score would change roundings and, eventually, visit counts.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/azhip.h"
#include "az_games.h"
#include "az_net_dev.h"
#include "az_net_hx_dev.h"
#include "az_host.h"

#define WAVES_PER_BLOCK 4
#define BLOCK_THREADS (64 * WAVES_PER_BLOCK)
#define PATH_LEN 48
#define HIST_PAD (AZ_MAX_MOVES + 6)

static thread_local std::string g_err;
int az_fail(const char* what, hipError_t e) {
    g_err = what;
    if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
    return -1;
}
#define HIPCHK(x) AZ_HIPCHK(x)

// ---------------------------------------------------------------------------
// device-side state
// ---------------------------------------------------------------------------
struct __attribute__((aligned(128))) Slot {
    // hot header, 64 B: every simulation kernel reads it with four 16-byte loads
    uint64_t nx, pv;                 // root position
    uint64_t leaf_nx, leaf_pv;       // pending leaf position
    uint32_t root, alloc;
    int32_t  nm;
    uint32_t root_ns;                // sum N of the root: a prefetch hint, always validated
    int32_t  leaf_kind, leaf_depth;
    uint32_t err, move_sims;
    // counters, 64 B
    uint64_t st_sims, st_steps, st_evals, st_root_evals, st_moves, st_games;
    uint64_t uid;
    int32_t  leaf_nm, first_version; // model version when this game started
    uint32_t path[PATH_LEN];         // (node << 4) | slot per select level
    uint8_t  hist[HIST_PAD];
    uint8_t  nedges[HIST_PAD];
};
static_assert(offsetof(Slot, st_sims) == 64 && offsetof(Slot, path) == 128, "Slot layout");

struct Dev {
    uint4*          pool;       // [G][cap][A+1]
    Slot*           slots;      // [G]
    uint32_t*       rec_visits; // [G][AZ_MAX_MOVES][AZ_MAX_ACTIONS]
    const double2*  lut;        // {Cs, sqrt(Ns)} by Ns
    az_game_record* ring;
    uint32_t*       ring_count;
    uint32_t        ring_cap;
    int32_t         lut_len;
    int32_t         G, cap, S, explore_depth, reuse, version;
    uint4*          cache;      // evaluation cache: 4 x 16 B per entry, or null
    uint32_t        cache_mask; // entries - 1
    unsigned long long* cache_hits;   // one counter per wavefront of the search grid (no atomics)
    int32_t*        order;      // one-launch-per-move kernel: [G + 64] game of each workgroup slot, then [G] evaluation
                                // counter of each game when its last such launch started
    double          alpha, eps, c_base, c_init;
    float           one_minus_eps;
    uint64_t        seed, uid_stride, first_uid;
};

struct az_engine {
    az_engine_config cfg;
    Dev d;
    int A;
    int num_cus;        // compute units of cfg.device
    int fused_wgs;      // workgroups az_c4_move_fused may use (0 = one per compute unit)
    uint32_t err_flags; // error bits seen by the last az_drain_finished
    double2* lut_dev;
};

// Every entry point that takes an engine runs with the engine's device current and puts the caller's device
// back on return: kernels are launched on the device that owns the pools, whatever device the calling thread
// had selected (the stream argument must belong to the engine's device).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};
#define ENGINE_SCOPE(e)                                                        \
    if (!(e)) return az_fail("null engine");                                   \
    DeviceScope _scope((e)->cfg.device);                                       \
    if (_scope.err != hipSuccess) return az_fail("selecting the engine's device", _scope.err)

// Diagnostic builds (-DAZ_EXP_ASSERT): every index that addresses a node pool, a slot, a queue or a plane is
// checked before use.  The first violation is recorded in the engine's counter block (words 24..27: code, value,
// workgroup, thread), error bit 6 is raised, and the access is skipped, so that a wrong index names itself
// instead of ending the launch with a memory fault.  Product builds compile the checks away.
#ifdef AZ_EXP_ASSERT
#define AZ_IDX_OK(E_, ok, code, val) az_idx_ok((E_), (ok), (code), (uint32_t)(val))
#else
#define AZ_IDX_OK(E_, ok, code, val) true
#endif

// ---------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint32_t uniu(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// max over each row of 16 lanes (butterfly: xor 1, xor 2, half-row mirror, row mirror)
__device__ __forceinline__ double row16_max(double x) {
#if AZ_USE_SHFL
    x = fmax(x, __shfl_xor(x, 1)); x = fmax(x, __shfl_xor(x, 2));
    x = fmax(x, __shfl_xor(x, 4)); x = fmax(x, __shfl_xor(x, 8));
#else
    x = fmax(x, dpp_f64<0xB1>(x));   // quad_perm [1,0,3,2]
    x = fmax(x, dpp_f64<0x4E>(x));   // quad_perm [2,3,0,1]
    x = fmax(x, dpp_f64<0x141>(x));  // row_half_mirror
    x = fmax(x, dpp_f64<0x140>(x));  // row_mirror
#endif
    return x;
}
__device__ __forceinline__ double row16_sum(double x) {
    x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
    return x;
}

// ---------------------------------------------------------------------------
// evaluation cache: (position, model version) -> the evaluator's (p, v), direct-mapped, 64 B per
// entry as four 16-byte words held by lanes 0..3 of a group:
//   0: {nx, pv}   1: {version, v, p0, p1}   2: {p2..p5}   3: {p6, p7, p8, check}
// Entries are written and read without locks; `check` folds the other fifteen words, so an entry
// torn between two writers reads as a miss.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t cache_slot(uint64_t nx, uint64_t pv) {
    uint32_t h = (uint32_t)nx * 0x9E3779B1u ^ rotl32((uint32_t)(nx >> 32) * 0x85EBCA77u, 13) ^
                 rotl32((uint32_t)pv * 0xC2B2AE3Du, 7) ^ (uint32_t)(pv >> 32) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t entry_fold(uint4 q, int sub) {
    uint32_t h = (q.x + 0x9E3779B9u * (uint32_t)(sub + 1)) * 0x85EBCA6Bu;
    h = (h ^ (h >> 15)) + q.y; h *= 0xC2B2AE35u;
    h = (h ^ (h >> 13)) + q.z; h *= 0x27D4EB2Fu;
    h = (h ^ (h >> 16)) + q.w; h *= 0x165667B1u;
    return h ^ (h >> 15);
}
// xor over the four lanes of a quad
__device__ __forceinline__ uint32_t quad_xor(uint32_t x) {
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    return x;
}

// alphazero/mcts.py:43-59 for one edge
__device__ __forceinline__ double puct_score(uint32_t N, float W, float P, double cs, double sq) {
    const float Q = N ? (float)((double)W / (double)N) : 0.0f;
    const double s = sq / (double)(N + 1u);
    const double U = (cs * (double)P) * s;
    return (double)Q + U;
}

#ifdef AZ_EXP_ASSERT
__device__ __forceinline__ bool az_idx_ok(const Dev& E, bool ok, uint32_t code, uint32_t val) {
    if (!ok) {
        uint32_t* dbg = E.ring_count + 24;
        if (atomicCAS(dbg, 0u, code) == 0u) { dbg[1] = val; dbg[2] = blockIdx.x; dbg[3] = threadIdx.x; }
        atomicOr(E.ring_count + 1, 64u);
    }
    return ok;
}
#endif
#ifdef AZ_EXP_LUT_CALL
// diagnostic builds only: the out-of-line call that the search step used to contain (never taken when the table
// covers the search), to compare spilling builds with and without a call in the kernel
__device__ __noinline__ double2 lut_slow(double c_base, double c_init, uint32_t Ns) {
    return make_double2(log((double)(1ull + Ns + (uint64_t)c_base) / c_base) + c_init, sqrt((double)Ns));
}
#endif
// {Cs, sqrt(Ns)} of a node whose edges have Ns visits in all.  The table is the host's (az_set_puct_lut: the
// reference's own numpy expressions), and a search never leaves it when it is sized as az_engine_create sizes it.
// A visit sum beyond its last row takes the last row and raises error bit 3 in the engine's flag word: the device
// has no logarithm that is bit-pinned to numpy's, so the search goes on but is reported as no longer exact.
__device__ __forceinline__ double2 lut_fetch(const Dev& E, uint32_t Ns) {
#ifdef AZ_EXP_LUT_CALL
    if ((int32_t)Ns >= 0 && (int32_t)Ns < E.lut_len) return E.lut[Ns];
    return lut_slow(E.c_base, E.c_init, Ns);
#else
    if (Ns >= (uint32_t)E.lut_len) {
        atomicOr(E.ring_count + 1, 8u);
        Ns = (uint32_t)E.lut_len - 1u;
    }
    return E.lut[Ns];
#endif
}
__device__ __forceinline__ void lut_lookup(const Dev& E, uint32_t Ns, double& cs, double& sq) {
    const double2 t = lut_fetch(E, Ns);
    cs = t.x; sq = t.y;
}

// ---------------------------------------------------------------------------
// counter-based RNG (Philox4x32-10) for production noise / move sampling
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                       uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
// two uniforms in [0,1) for (uid, move, purpose, attempt)
__device__ __forceinline__ void rng2(const Dev& E, uint64_t uid, uint32_t move, uint32_t purpose,
                                     uint32_t attempt, double& a, double& b) {
    uint32_t o[4];
    philox(attempt, (move << 8) | purpose, (uint32_t)uid, (uint32_t)(uid >> 32),
           (uint32_t)E.seed, (uint32_t)(E.seed >> 32), o);
    a = u53(o[0], o[1]); b = u53(o[2], o[3]);
}
// Gamma(alpha, 1) by Marsaglia-Tsang (alpha < 1 boosted); purpose = edge index
__device__ double gamma_draw(const Dev& E, uint64_t uid, uint32_t move, uint32_t edge, double alpha) {
    const double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (uint32_t t = 0; t < 64; ++t) {
        double u1, u2, u3, u4;
        rng2(E, uid, move, edge, 2 * t + 2, u1, u2);
        rng2(E, uid, move, edge, 2 * t + 3, u3, u4);
        const double x = sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double uu = 1.0 - u3;   // in (0, 1]
        if (uu < 1.0 - 0.0331 * x * x * x * x || log(uu) < 0.5 * x * x + d * (1.0 - v + log(v))) {
            g = d * v;
            break;
        }
    }
    if (alpha < 1.0) {
        double u1, u2;
        rng2(E, uid, move, edge, 1, u1, u2);
        g *= pow(1.0 - u1, 1.0 / alpha);
    }
    return g;
}

// ---------------------------------------------------------------------------
// per-tree primitives of the once-per-move kernels (one wavefront per tree)
// ---------------------------------------------------------------------------
// mcts.py:20-26: one new node; priors compacted to the legal moves in ascending
// action order.  p == nullptr gives the UniformModel prior 1/n (models.py:90-95).
template <class GM>
__device__ uint32_t expand_node(const Dev& E, uint4* pool, uint32_t& alloc, uint64_t nx, uint64_t pv,
                                const float* p, uint32_t& err) {
    constexpr int AS = GM::A + 1;
    const int lane = lane_id();
    if (alloc >= (uint32_t)E.cap) { err |= 1u; return AZ_NO_NODE; }
    const uint32_t idx = alloc;
    alloc += 1;
    const uint32_t legal = GM::legal(nx, pv);
    const int n_edges = az_popc32(legal);
    if (lane < AS) {
        uint4 rec = make_uint4(0u, 0u, 0u, AZ_NO_NODE);
        if (lane < n_edges) {
            const int a = az_nth_set_bit(legal, lane);
            const float pr = p ? p[a] : 1.0f / (float)n_edges;
            rec.z = __float_as_uint(pr);
        }
        if (lane == AS - 1) rec = make_uint4((uint32_t)n_edges, 0u, 0u, 0u);
        pool[(size_t)idx * AS + lane] = rec;
    }
    return idx;
}

// mcts.py:88-91
template <class GM>
__device__ void root_noise(const Dev& E, uint4* pool, uint32_t root, const double* eta_g, uint64_t uid,
                           int move) {
    constexpr int AS = GM::A + 1;
    const int lane = lane_id();
    const int n_edges = (int)pool[(size_t)root * AS + AS - 1].x;
    double eta = 0.0;
    if (eta_g) {
        if (lane < n_edges) eta = eta_g[lane];
    } else {
        double g = 0.0;
        if (lane < n_edges) g = gamma_draw(E, uid, (uint32_t)move, (uint32_t)lane, E.alpha);
        const double s = row16_sum(g);
        eta = s > 0.0 ? g / s : 1.0 / (double)n_edges;
    }
    if (lane < n_edges) {
        uint32_t* pp = reinterpret_cast<uint32_t*>(&pool[(size_t)root * AS + lane]) + 2;
        const float a = E.one_minus_eps * __uint_as_float(*pp);
        *pp = __float_as_uint((float)((double)a + E.eps * eta));
    }
}

// ---------------------------------------------------------------------------
// kernels: one wave per game
// ---------------------------------------------------------------------------
#define WAVE_PROLOGUE                                                                   \
    const int g = uni((int)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)));       \
    if (g >= E.G) return;                                                               \
    const int lane = lane_id();                                                         \
    Slot* s = E.slots + g;                                                              \
    uint4* pool = E.pool + (size_t)g * E.cap * (GM::A + 1);                             \
    (void)lane; (void)pool;

template <class GM>
__global__ void __launch_bounds__(BLOCK_THREADS) k_reset(Dev E) {
    WAVE_PROLOGUE
    if (lane == 0) {
        s->nx = 0; s->pv = 0; s->nm = 0;
        s->uid = E.first_uid + (uint64_t)g;
        s->root = AZ_NO_NODE; s->alloc = 0;
        s->leaf_kind = AZ_LEAF_NONE; s->leaf_depth = 0; s->leaf_nm = 0; s->leaf_nx = 0; s->leaf_pv = 0;
        s->err = 0; s->move_sims = 0; s->first_version = E.version;
        s->st_sims = s->st_steps = s->st_evals = s->st_root_evals = s->st_moves = s->st_games = 0;
    }
}

// every game in progress is given up and its slot starts the game that would have come next (the uid moves on, so no
// random stream is used twice); statistics, error bits and the record ring stay as they are
template <class GM>
__global__ void __launch_bounds__(BLOCK_THREADS) k_abandon(Dev E) {
    WAVE_PROLOGUE
    if (lane == 0) {
        s->nx = 0; s->pv = 0; s->nm = 0;
        s->uid += E.uid_stride;
        s->root = AZ_NO_NODE; s->alloc = 0;
        s->leaf_kind = AZ_LEAF_NONE; s->leaf_depth = 0; s->leaf_nm = 0; s->leaf_nx = 0; s->leaf_pv = 0;
        s->move_sims = 0; s->first_version = E.version;
    }
}

// mcts.py:114-117.  UNIFORM: expand with 1/n in place and add the noise here.
template <class GM, bool UNIFORM>
__global__ void __launch_bounds__(BLOCK_THREADS)
k_begin_move(Dev E, uint64_t* leaf_bb, int32_t* leaf_info, const double* eta) {
    WAVE_PROLOGUE
    uint32_t root = s->root, alloc = s->alloc, err = s->err;
    const uint64_t nx = s->nx, pv = s->pv;
    const int nm = s->nm;
    const bool need = (root == AZ_NO_NODE) && GM::winner(nx, pv, nm) == AZ_W_NONE;
    if (UNIFORM) {
        if (need) {
            root = expand_node<GM>(E, pool, alloc, nx, pv, nullptr, err);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
        if (root != AZ_NO_NODE)
            root_noise<GM>(E, pool, root, eta ? eta + (size_t)g * GM::A : nullptr, s->uid, nm);
        if (lane == 0) {
            s->root = root; s->alloc = alloc; s->err = err; s->move_sims = 0;
            s->leaf_kind = AZ_LEAF_NONE;
            if (need) s->st_root_evals += 1;
        }
    } else {
        if (lane == 0) {
            s->leaf_kind = need ? AZ_LEAF_ROOT : AZ_LEAF_NONE;
            s->leaf_depth = 0; s->leaf_nx = nx; s->leaf_pv = pv; s->leaf_nm = nm;
            leaf_bb[2 * (size_t)g] = nx; leaf_bb[2 * (size_t)g + 1] = pv;
            leaf_info[g] = (need ? AZ_LEAF_ROOT : AZ_LEAF_NONE) | (nm << 8) | ((int)GM::legal(nx, pv) << 16);
        }
    }
}

template <class GM>
__global__ void __launch_bounds__(BLOCK_THREADS) k_apply_root(Dev E, const float* p, const double* eta) {
    WAVE_PROLOGUE
    uint32_t root = s->root, alloc = s->alloc, err = s->err;
    const bool need = s->leaf_kind == AZ_LEAF_ROOT;
    if (need) {
        root = expand_node<GM>(E, pool, alloc, s->nx, s->pv, p + (size_t)g * GM::A, err);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    if (root != AZ_NO_NODE)
        root_noise<GM>(E, pool, root, eta ? eta + (size_t)g * GM::A : nullptr, s->uid, s->nm);
    if (lane == 0) {
        s->root = root; s->alloc = alloc; s->err = err; s->move_sims = 0;
        s->leaf_kind = AZ_LEAF_NONE;
        if (need) s->st_root_evals += 1;
    }
}

// selfplay.py:26-38 (+ :41 for finished games)
template <class GM>
__global__ void __launch_bounds__(BLOCK_THREADS)
k_choose(Dev E, const double* u_in, int32_t* out_action, int32_t* out_nedges, uint32_t* out_visits,
         int32_t* out_winner) {
    WAVE_PROLOGUE
    constexpr int AS = GM::A + 1;
    const uint32_t root = s->root;
    // pool exhaustion (bit 0) / record ring overflow (bit 1) of any game since the last reset: one word next
    // to the ring counter, fetched by az_drain_finished together with it
    if (lane == 0 && s->err) atomicOr(E.ring_count + 1, s->err);
    uint64_t nx = s->nx, pv = s->pv;
    int nm = s->nm;
    if (root == AZ_NO_NODE) {   // nothing searched (terminal or never begun): report and leave
        if (lane == 0) {
            if (out_action) out_action[g] = -1;
            if (out_nedges) out_nedges[g] = 0;
            if (out_winner) out_winner[g] = GM::winner(nx, pv, nm);
        }
        if (out_visits && lane < GM::A) out_visits[(size_t)g * GM::A + lane] = 0;
        return;
    }
    uint4 e = make_uint4(0u, 0u, 0u, AZ_NO_NODE);
    if (lane < AS) e = pool[(size_t)root * AS + lane];
    const int n_edges = (int)rdlane(e.x, AS - 1);
    const uint32_t N = lane < n_edges ? e.x : 0u;

    int slot;
    if (nm < E.explore_depth) {
        // mcts.py:61-71: pi = N/sum (fp64), legacy RandomState.choice = cdf search
        double u;
        if (u_in) u = u_in[g];
        else { double t; rng2(E, s->uid, (uint32_t)nm, 0xFFu, 0, u, t); }
        double ssum = 0.0;
        for (int i = 0; i < n_edges; ++i) ssum += (double)rdlane(N, i);
        double cdf[AZ_MAX_ACTIONS];
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < GM::A; ++i) {
            if (i < n_edges) {
                const double pi = ssum > 0.0 ? (double)rdlane(N, i) / ssum : 1.0 / (double)n_edges;
                acc += pi;
            }
            cdf[i] = acc;
        }
        const double last = acc;
        int idx = 0;
#pragma unroll
        for (int i = 0; i < GM::A; ++i)
            if (i < n_edges && cdf[i] / last <= u) idx = i + 1;
        slot = idx < n_edges ? idx : n_edges - 1;
    } else {
        // mcts.py:73-74: first maximum of N
        uint32_t best = rdlane(N, 0);
        slot = 0;
        for (int i = 1; i < n_edges; ++i) {
            const uint32_t ni = rdlane(N, i);
            if (ni > best) { best = ni; slot = i; }
        }
    }
    slot = uni(slot);
    const int action = az_nth_set_bit(GM::legal(nx, pv), slot);
    const uint32_t child = rdlane(e.w, slot);

    // selfplay.py:31: search statistics of this move
    if (lane < AZ_MAX_ACTIONS)
        E.rec_visits[((size_t)g * AZ_MAX_MOVES + nm) * AZ_MAX_ACTIONS + lane] = N;
    if (out_visits && lane < GM::A) out_visits[(size_t)g * GM::A + lane] = N;
    if (lane == 0) {
        s->hist[nm] = (uint8_t)action;
        s->nedges[nm] = (uint8_t)n_edges;
        if (out_action) out_action[g] = action;
        if (out_nedges) out_nedges[g] = n_edges;
    }
    GM::play(nx, pv, nm, action);
    const int w = GM::winner(nx, pv, nm);
    if (lane == 0 && out_winner) out_winner[g] = w;

    if (w != AZ_W_NONE) {
        // selfplay.py:41: hand the finished game over, then restart the slot (selfplay.py:13-14)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(E.ring_count, 1u);
        pos = uniu(pos);
        if (pos < E.ring_cap) {
            az_game_record* r = E.ring + pos;
            if (lane == 0) {
                r->uid = s->uid; r->num_moves = nm; r->winner = w; r->model_version = E.version;
                r->first_model_version = s->first_version;
            }
            if (lane < HIST_PAD) { r->history[lane] = s->hist[lane]; r->num_edges[lane] = s->nedges[lane]; }
            const uint32_t* src = E.rec_visits + (size_t)g * AZ_MAX_MOVES * AZ_MAX_ACTIONS;
            uint32_t* dst = &r->visits[0][0];
            for (int i = lane; i < AZ_MAX_MOVES * AZ_MAX_ACTIONS; i += 64)
                dst[i] = (i < nm * AZ_MAX_ACTIONS) ? src[i] : 0u;
        }
        if (lane == 0) {
            if (pos >= E.ring_cap) { s->err |= 2u; atomicOr(E.ring_count + 1, 2u); }
            s->first_version = E.version;
            s->nx = 0; s->pv = 0; s->nm = 0; s->root = AZ_NO_NODE; s->alloc = 0;
            s->uid += E.uid_stride;
            s->st_moves += 1; s->st_games += 1; s->leaf_kind = AZ_LEAF_NONE; s->move_sims = 0;
        }
    } else if (lane == 0) {
        s->nx = nx; s->pv = pv; s->nm = nm;
        s->root_ns = rdlane(N, slot) - 1u;                  // prefetch hint for the new root's sum N
        if (E.reuse) s->root = child;                       // selfplay.py:35-36
        else { s->root = AZ_NO_NODE; s->alloc = 0; }        // selfplay.py:38
        s->st_moves += 1; s->leaf_kind = AZ_LEAF_NONE; s->move_sims = 0;
    }
}

// ---------------------------------------------------------------------------
// packed search kernels: GS lanes per game (8 for Connect Four, 16 for tic-tac-toe), so a
// wavefront runs 64/GS trees at once.  A node's A+1 records map onto the group's lanes, the
// arg-max is a DPP butterfly inside the group, and every "uniform per tree" value lives
// replicated in the group's lanes.  Trees of one wave descend in lock-step; a group whose
// simulation has reached its leaf idles (exec-masked) until the deepest one is done.  Compared
// with a whole wavefront per tree this issues 64/GS times fewer vector instructions, which is
// what bounded the one-tree-per-wave form (its memory latency was already hidden by 8 waves
// per SIMD).
// ---------------------------------------------------------------------------
template <int GS>
struct Grp {
    static constexpr int GPW = 64 / GS;    // games per wave
    __device__ static __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, GS); }
    __device__ static __forceinline__ double gmax(double x) {
        x = fmax(x, dpp_f64<0xB1>(x));    // quad_perm [1,0,3,2]
        x = fmax(x, dpp_f64<0x4E>(x));    // quad_perm [2,3,0,1]
        x = fmax(x, dpp_f64<0x141>(x));   // row_half_mirror
        if (GS >= 16) x = fmax(x, dpp_f64<0x140>(x));  // row_mirror
        return x;
    }
    // predicate bits of this lane's group, bit i = lane (group base + i), first 16 lanes only
    __device__ static __forceinline__ uint32_t gballot(bool p) {
        const unsigned long long m = __ballot(p);
        return (uint32_t)(m >> (lane_id() & ~(GS - 1))) & (GS >= 16 ? 0xFFFFu : ((1u << GS) - 1u));
    }
};

// mcts.py:119-134 for the group's tree; path[] is this game's row of the block's LDS path stack.
// ns_hint: expected sum N of the root.  The {Cs, sqrt(Ns)} entry of a node is fetched together
// with the node itself, for the sum N the node is expected to have (a child reached through an
// edge with N visits was expanded on the first of them, so its edges sum to N-1); the entry is
// re-fetched if the node's own sum differs, so the hint never affects results.
template <class GM, int GS>
__device__ __forceinline__ void select_leaf_g(const Dev& E, const uint4* pool, uint32_t root, uint64_t& nx,
                                              uint64_t& pv, int& nm, uint32_t* path, int sub, int& kind,
                                              int& depth, float& vterm, uint32_t& steps, uint32_t& ns_hint,
                                              bool probe, uint4& cq) {
    constexpr int AS = GM::A + 1;
    static_assert(AS <= GS, "group too small for this game");
    using G_ = Grp<GS>;
    kind = AZ_LEAF_NONE; depth = 0; vterm = 0.0f;
    uint32_t node = root;
    bool active = root != AZ_NO_NODE && GM::winner(nx, pv, nm) == AZ_W_NONE;
    uint32_t hint = ns_hint;
    double2 lt = make_double2(0.0, 0.0);
    if (active) lt = lut_fetch(E, hint);
    uint4 e = make_uint4(0u, 0u, 0u, AZ_NO_NODE);
    if (active && !AZ_IDX_OK(E, node < (uint32_t)E.cap, 3, node)) active = false;
    if (active && sub < AS) e = pool[(size_t)node * AS + sub];
    while (active) {
        const int n_edges = (int)G_::bcast(e.x, AS - 1);
        const uint32_t Ns = G_::bcast(e.y, AS - 1);
        if (Ns != hint) lt = lut_fetch(E, Ns);
        if (depth == 0) ns_hint = Ns + 1u;                             // this simulation adds one root visit
        const bool act = sub < n_edges;
        double score = -INFINITY;
        if (act) score = puct_score(e.x, __uint_as_float(e.y), __uint_as_float(e.z), lt.x, lt.y);
        const double mx = G_::gmax(score);
        const uint32_t hit = G_::gballot(act && score == mx);
        const int slot = __ffs((int)hit) - 1;                          // np.argmax: first maximum
        const uint32_t child = G_::bcast(e.w, slot);
        hint = G_::bcast(e.x, slot) - 1u;
        // the child's line and its table entry are requested here, so that stepping the position and
        // the win test below run while they travel
        uint4 e_next = make_uint4(0u, 0u, 0u, AZ_NO_NODE);
        if (child != AZ_NO_NODE && AZ_IDX_OK(E, child < (uint32_t)E.cap, 1, child)) {
            lt = lut_fetch(E, hint);
            if (sub < AS) e_next = pool[(size_t)child * AS + sub];
        }
        const int action = az_nth_set_bit(GM::legal(nx, pv), slot);    // mcts.py:126
        if (sub == 0 && AZ_IDX_OK(E, depth < PATH_LEN && slot >= 0 && slot < AS, 2, (depth << 8) | (slot & 255)))
            path[depth] = (node << 4) | (uint32_t)slot;
        depth += 1;
        steps += 1;
        GM::play(nx, pv, nm, action);
        const int w = GM::winner(nx, pv, nm);
        if (w != AZ_W_NONE) {                                          // mcts.py:131-134
            kind = AZ_LEAF_TERMINAL;
            vterm = az_terminal_value_prev(w, nm);
            active = false;
        } else if (child == AZ_NO_NODE) {                              // mcts.py:135-138
            kind = AZ_LEAF_EVAL;
            active = false;
            // the cache line travels while the other trees of the wave finish their descent
            if (probe && sub < 4) cq = E.cache[(size_t)(cache_slot(nx, pv) & E.cache_mask) * 4 + sub];
        } else {
            node = child;
        }
        e = e_next;
    }
}

template <class GM, int GS>
__device__ __forceinline__ uint32_t expand_node_g(const Dev& E, uint4* pool, uint32_t& alloc, uint64_t nx,
                                                  uint64_t pv, const float* p, uint32_t& err, int sub) {
    constexpr int AS = GM::A + 1;
    if (alloc >= (uint32_t)E.cap) { err |= 1u; return AZ_NO_NODE; }
    const uint32_t idx = alloc;
    alloc += 1;
    const uint32_t legal = GM::legal(nx, pv);
    const int n_edges = az_popc32(legal);
    if (sub < AS) {
        uint4 rec = make_uint4(0u, 0u, 0u, AZ_NO_NODE);
        if (sub < n_edges) {
            const int a = az_nth_set_bit(legal, sub);
            const float pr = p ? p[a] : 1.0f / (float)n_edges;
            rec.z = __float_as_uint(pr);
        }
        if (sub == AS - 1) rec = make_uint4((uint32_t)n_edges, 0u, 0u, 0u);
        pool[(size_t)idx * AS + sub] = rec;
    }
    return idx;
}

// mcts.py:84-86 for one path level: W += v (fp32), N += 1, and the node's cached sum N
template <class GM>
__device__ __forceinline__ void backup_edge(const Dev& E, uint4* pool, uint32_t ent, float v) {
    constexpr int AS = GM::A + 1;
    const uint32_t node = ent >> 4, slot = ent & 15u;
    if (!AZ_IDX_OK(E, node < (uint32_t)E.cap && slot < (uint32_t)AS, 4, ent)) return;
    uint2* ep = reinterpret_cast<uint2*>(&pool[(size_t)node * AS + slot]);
    uint32_t* ns = reinterpret_cast<uint32_t*>(&pool[(size_t)node * AS + AS - 1]) + 1;
    uint2 nw = *ep;                       // both loads are issued before either store
    const uint32_t nsv = *ns;
    nw.x += 1u;
    nw.y = __float_as_uint(__uint_as_float(nw.y) + v);
    *ep = nw;
    *ns = nsv + 1u;
}
// mcts.py:147-158: the group's lanes take path levels sub, sub+GS, ... (levels are distinct nodes)
template <class GM, int GS>
__device__ __forceinline__ void backup_g(const Dev& E, uint4* pool, int depth, const uint32_t* path, float v_leaf_edge, int sub) {
    for (int d = sub; d < depth; d += GS)
        backup_edge<GM>(E, pool, path[d], ((depth - 1 - d) & 1) ? -v_leaf_edge : v_leaf_edge);
}

#define GROUP_PROLOGUE                                                                   \
    constexpr int GPW = 64 / GS;                                                         \
    __shared__ uint32_t lds_path[WAVES_PER_BLOCK * GPW][PATH_LEN];                       \
    const int lane = lane_id(), sub = lane & (GS - 1);                                   \
    const int gl = (int)(threadIdx.x >> 6) * GPW + lane / GS;                            \
    const int g = (int)blockIdx.x * (WAVES_PER_BLOCK * GPW) + gl;                        \
    if (g >= E.G) return;                                                                \
    Slot* s = E.slots + g;                                                               \
    uint4* pool = E.pool + (size_t)g * E.cap * (GM::A + 1);                              \
    uint32_t* path = lds_path[gl];

// mcts.py:119-158, one simulation per game per launch, GS lanes per game:
//   DO_EXPAND: finish the pending simulation with the evaluator's (p, v);
//   DO_SELECT: start the next one; terminal leaves are backed up at once.
// Everything the kernel needs from the slot, the saved path and (p, v) is requested up front in one
// batch of loads, so the dependent chain is: slot -> edges to back up -> root -> one line per level.
// `path` is the group's scratch for (node, slot) per level: a row of an LDS stack, or the slot's own
// path array in global memory (PATH_IN_SLOT), in which case nothing has to be copied for the next launch.
// `sel` (when DO_SELECT): a run-time switch, the same for all lanes of a group, that turns this call's select
// off.  Returns the kind of the leaf the call ends with (AZ_LEAF_NONE without a select).
template <class GM, int GS, bool DO_EXPAND, bool DO_SELECT, bool PATH_IN_SLOT = false>
__device__ __forceinline__ int sim_step_game(const Dev& E, const float* p, const float* v, uint64_t* leaf_bb,
                                             int32_t* leaf_info, int g, int lane, int sub, Slot* s, uint4* pool,
                                             uint32_t* path, int wave_id, uint64_t* lds_leaf = nullptr, bool sel = true) {
    using G_ = Grp<GS>;
    constexpr int AS = GM::A + 1;
    const uint4* sh = reinterpret_cast<const uint4*>(s);
    const uint4 h0 = sh[0], h1 = sh[1], h2 = sh[2], h3 = sh[3];
    const uint4 c0 = sh[4], c1 = sh[5];                              // st_sims, st_steps | st_evals, st_root_evals
    uint32_t pa = 0, pb = 0;
    float p_lane = 0.0f, v_g = 0.0f;
    if (DO_EXPAND) {
        pa = s->path[sub];                                           // levels 0..GS-1 and GS..2GS-1
        if (GS < PATH_LEN && sub + GS < PATH_LEN) pb = s->path[sub + GS];
        if (sub < GM::A) p_lane = p[(size_t)g * GM::A + sub];
        v_g = v[g];
    }
    uint64_t nx = (uint64_t)h0.x | ((uint64_t)h0.y << 32), pv = (uint64_t)h0.z | ((uint64_t)h0.w << 32);
    const uint64_t leaf_nx = (uint64_t)h1.x | ((uint64_t)h1.y << 32), leaf_pv = (uint64_t)h1.z | ((uint64_t)h1.w << 32);
    const uint32_t root = h2.x;
    uint32_t alloc = h2.y, root_ns = h2.w, err = h3.z;
    int nm = (int)h2.z, kind = (int)h3.x;
    uint32_t sims = 0, evals = 0, steps = 0;

    if (DO_EXPAND && (kind == AZ_LEAF_EVAL || kind == AZ_LEAF_CACHED)) {
        int depth = (int)h3.y;
        if (!AZ_IDX_OK(E, depth >= 1 && depth <= PATH_LEN, 5, depth)) depth = 1;
        if (E.cache && kind == AZ_LEAF_EVAL) {                       // remember what the evaluator said
            const int w0 = 4 * sub - 6;                              // index into p of this lane's word .x
            uint32_t w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int a = w0 + c;
                const uint32_t t = G_::bcast(__float_as_uint(p_lane), a & (GS - 1));
                w[c] = (a >= 0 && a < GM::A) ? t : 0u;
            }
            uint4 q = make_uint4(w[0], w[1], w[2], w[3]);
            if (sub == 0) q = make_uint4((uint32_t)leaf_nx, (uint32_t)(leaf_nx >> 32), (uint32_t)leaf_pv, (uint32_t)(leaf_pv >> 32));
            if (sub == 1) { q.x = (uint32_t)E.version; q.y = __float_as_uint(v_g); }
            const uint32_t chk = quad_xor(sub < 4 ? entry_fold(q, sub) : 0u);
            if (sub == 3) q.w = chk;
            if (sub < 4) E.cache[(size_t)(cache_slot(leaf_nx, leaf_pv) & E.cache_mask) * 4 + sub] = q;
        }
        // mcts.py:20-26: the new node, priors compacted to the legal moves
        uint32_t idx = AZ_NO_NODE;
        if (alloc < (uint32_t)E.cap) { idx = alloc; alloc += 1; } else err |= 1u;
        const uint32_t legal = GM::legal(leaf_nx, leaf_pv);
        const int n_edges = az_popc32(legal);
        const int my_action = sub < n_edges ? az_nth_set_bit(legal, sub) : 0;
        const float pr = __uint_as_float(G_::bcast(__float_as_uint(p_lane), my_action));
        if (idx != AZ_NO_NODE && sub < AS) {
            uint4 rec = make_uint4(0u, 0u, sub < n_edges ? __float_as_uint(pr) : 0u, AZ_NO_NODE);
            if (sub == AS - 1) rec = make_uint4((uint32_t)n_edges, 0u, 0u, 0u);
            pool[(size_t)idx * AS + sub] = rec;
        }
        const int dl = depth - 1;
        const uint32_t last = G_::bcast(dl < GS ? pa : pb, dl & (GS - 1));
        if (2 * GS < PATH_LEN && depth > 2 * GS) { /* deeper than the two registers hold: rare, read it */
            const uint32_t l2 = s->path[dl];
            if (sub == 0 && idx != AZ_NO_NODE)
                reinterpret_cast<uint32_t*>(&pool[(size_t)(l2 >> 4) * AS + (l2 & 15u)])[3] = idx;
        } else if (idx != AZ_NO_NODE && sub == 0 && AZ_IDX_OK(E, (last >> 4) < (uint32_t)E.cap && (last & 15u) < (uint32_t)AS, 6, last)) {
            reinterpret_cast<uint32_t*>(&pool[(size_t)(last >> 4) * AS + (last & 15u)])[3] = idx;
        }
        // mcts.py:144-158: v *= -1, then one level per lane
        const float vb = -v_g;
        if (sub < depth) backup_edge<GM>(E, pool, pa, ((depth - 1 - sub) & 1) ? -vb : vb);
        if (GS < PATH_LEN && sub + GS < depth) backup_edge<GM>(E, pool, pb, ((depth - 1 - sub - GS) & 1) ? -vb : vb);
        for (int d = sub + 2 * GS; d < depth; d += GS)
            backup_edge<GM>(E, pool, s->path[d], ((depth - 1 - d) & 1) ? -vb : vb);
        sims += 1; evals += 1;
        kind = AZ_LEAF_NONE;
    }
    if (DO_EXPAND && DO_SELECT && sel) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // own stores before re-reading nodes

    int depth = 0;
    if (DO_SELECT && sel) {
        float vterm;
        uint4 cq = make_uint4(0u, 0u, 0u, 0u);
        const bool probe = DO_EXPAND && E.cache != nullptr;
        select_leaf_g<GM, GS>(E, pool, root, nx, pv, nm, path, sub, kind, depth, vterm, steps, root_ns, probe, cq);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // path[] is written by lane 0 of the group
        if (DO_EXPAND && probe) {
            bool hit = false;
            if (kind == AZ_LEAF_EVAL) {
                uint4 f = cq;
                if (sub == 3) f.w = 0u;
                const uint32_t chk = quad_xor(sub < 4 ? entry_fold(f, sub) : 0u);
                bool ok = true;
                if (sub == 0) ok = cq.x == (uint32_t)nx && cq.y == (uint32_t)(nx >> 32) && cq.z == (uint32_t)pv && cq.w == (uint32_t)(pv >> 32);
                if (sub == 1) ok = cq.x == (uint32_t)E.version;
                if (sub == 3) ok = cq.w == chk;
                hit = (G_::gballot(ok) & 0xFu) == 0xFu;
                if (hit) {
                    // this position was evaluated before with these weights: the evaluator's row is
                    // filled in here and the evaluator skips it
                    const int wi = 6 + sub, src = (wi >> 2) & 3;
                    const uint32_t t0 = G_::bcast(cq.x, src), t1 = G_::bcast(cq.y, src);
                    const uint32_t t2 = G_::bcast(cq.z, src), t3 = G_::bcast(cq.w, src);
                    const uint32_t vb = G_::bcast(cq.y, 1);
                    const uint32_t pw = (wi & 2) ? ((wi & 1) ? t3 : t2) : ((wi & 1) ? t1 : t0);
                    if (sub < GM::A) const_cast<float*>(p)[(size_t)g * GM::A + sub] = __uint_as_float(pw);
                    if (sub == 0) const_cast<float*>(v)[g] = __uint_as_float(vb);
                    kind = AZ_LEAF_CACHED;
                }
            }
            const unsigned long long hm = __ballot(hit && sub == 0);
            if (hm && lane == __ffsll((long long)hm) - 1) {        // one of the active lanes
                unsigned long long* hc = E.cache_hits + wave_id;
                *hc += (unsigned long long)__popcll(hm);
            }
        }
        if (kind == AZ_LEAF_TERMINAL) {
            backup_g<GM, GS>(E, pool, depth, path, vterm, sub);
            sims += 1;
        } else if (!PATH_IN_SLOT && (kind == AZ_LEAF_EVAL || kind == AZ_LEAF_CACHED)) {
            for (int d = sub; d < depth; d += GS) s->path[d] = path[d];
        }
        if (sub == 0) {
            leaf_bb[2 * (size_t)g] = nx; leaf_bb[2 * (size_t)g + 1] = pv;
            leaf_info[g] = kind | (nm << 8) | ((int)GM::legal(nx, pv) << 16);
            if (lds_leaf) { lds_leaf[0] = nx; lds_leaf[1] = pv; }      // a network in the same launch reads it here
        }
    }
    if (sub == 0) {
        uint4* so = reinterpret_cast<uint4*>(s);
        if (DO_SELECT && sel) so[1] = make_uint4((uint32_t)nx, (uint32_t)(nx >> 32), (uint32_t)pv, (uint32_t)(pv >> 32));
        so[2] = make_uint4(root, alloc, h2.z, root_ns);
        so[3] = make_uint4((uint32_t)kind, (DO_SELECT && sel) ? (uint32_t)depth : h3.y, err, h3.w + sims);
        const uint64_t st_sims = ((uint64_t)c0.x | ((uint64_t)c0.y << 32)) + sims;
        const uint64_t st_steps = ((uint64_t)c0.z | ((uint64_t)c0.w << 32)) + steps;
        const uint64_t st_evals = ((uint64_t)c1.x | ((uint64_t)c1.y << 32)) + evals;
        so[4] = make_uint4((uint32_t)st_sims, (uint32_t)(st_sims >> 32), (uint32_t)st_steps, (uint32_t)(st_steps >> 32));
        so[5] = make_uint4((uint32_t)st_evals, (uint32_t)(st_evals >> 32), c1.z, c1.w);
        if (DO_SELECT && sel) s->leaf_nm = nm;
    }
    return kind;
}

template <class GM, int GS, bool DO_EXPAND, bool DO_SELECT>
__global__ void __launch_bounds__(BLOCK_THREADS)
k_sim_step_p(Dev E, const float* p, const float* v, uint64_t* leaf_bb, int32_t* leaf_info) {
    GROUP_PROLOGUE
    sim_step_game<GM, GS, DO_EXPAND, DO_SELECT>(E, p, v, leaf_bb, leaf_info, g, lane, sub, s, pool, path,
                                                (int)blockIdx.x * WAVES_PER_BLOCK + (int)(threadIdx.x >> 6));
}

// ---------------------------------------------------------------------------
// Which games each workgroup of the one-launch-per-move kernel serves.  A launch lasts as long as its slowest
// workgroup, and a game's cost per simulation (how many of its leaves need the network) changes slowly from one
// move to the next, so the games are dealt by their evaluation count of the previous such launch: sorted by it,
// then row by row across the workgroups, every other row backwards.  Games that have just started count as the
// most expensive.  One workgroup; a counting sort on keys 0..1023.  order[w * gpw + i] = i-th game of workgroup w,
// -1 behind the last one.  Which workgroup serves a game has no influence on its tree.
// ---------------------------------------------------------------------------
#define BALANCE_KEYS 1024
#define BALANCE_MAX_GAMES 16384
__global__ void __launch_bounds__(1024) k_fused_balance(Dev E, int gpw, int grid) {
    __shared__ int start[BALANCE_KEYS];
    __shared__ int scan[BALANCE_KEYS];
    __shared__ uint16_t keys[BALANCE_MAX_GAMES];
    const int tid = threadIdx.x;
    int32_t* order = E.order;
    uint32_t* last = reinterpret_cast<uint32_t*>(E.order + E.G + 64);
    start[tid] = 0;
    for (int i = tid; i < grid * gpw; i += 1024) order[i] = -1;
    __syncthreads();
    for (int g = tid; g < E.G; g += 1024) {
        const Slot* s = E.slots + g;
        const uint32_t ev = (uint32_t)s->st_evals;
        uint32_t key = ev - last[g];
        if (key > BALANCE_KEYS - 1 || s->nm == 0) key = BALANCE_KEYS - 1;
        last[g] = ev;
        keys[g] = (uint16_t)key;
        atomicAdd(&start[BALANCE_KEYS - 1 - key], 1);             // bins in descending key order
    }
    __syncthreads();
    // exclusive prefix sum of the bins
    const int mine = start[tid];
    scan[tid] = mine;
    __syncthreads();
    for (int d = 1; d < BALANCE_KEYS; d <<= 1) {
        const int add = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    start[tid] = scan[tid] - mine;
    __syncthreads();
    for (int g = tid; g < E.G; g += 1024) {
        const int r = atomicAdd(&start[BALANCE_KEYS - 1 - keys[g]], 1);
        const int row = r / grid;
        int w = r - row * grid;
        if (row & 1) w = grid - 1 - w;
        order[w * gpw + row] = g;
    }
}

// ---------------------------------------------------------------------------
// A whole move's simulations in ONE launch, Connect Four with the fused <= 32-channel network: a
// workgroup owns `gpw` games from the first select to the last backup.  A leaf is evaluated by the
// workgroup that owns its game, so there is no grid-wide synchronisation, no kernel boundary between
// search and network, and the network weights are fetched (into LDS) once per move instead of once per
// simulation.
//
// Inside the workgroup nothing runs in lock step either.  Every game goes round its own cycle
//     ready queue --a wave's search step: (expand/backup of the last evaluation,) select-->
//         leaf queue    (the leaf needs the network)
//         ready queue   (terminal leaf, backed up at once; or the evaluation was found in the cache)
//     leaf queue --a wave's network task: stem, tower and head convs of two queued leaves (one if no other is
//         in sight) in the wave's registers, head planes into the games' slots in LDS--> plane queue
//     plane queue --one wave: the fully connected heads of ALL boards that wait, up to 32 columns of the
//         same MFMAs--> (p, v), ready queue
// until the expand/backup that follows its num_sims-th select.  The FUSED_WAVES waves are identical workers: a
// worker takes a network task when two leaves are queued; else the heads, if eight boards wait for them (or any,
// when there is nothing to search) and no other wave is at it; else up to eight ready games for one search
// step (eight lanes per tree); else it sleeps.  So the tree walks' memory latency hides behind other games'
// matrix work, each SIMD has a second wave's MFMA stream to fill the gaps of the first most of the time, the
// heads (latency-bound chains that cost a wave the same for 32 boards as for two) are paid once per six
// boards on average, and no wave ever waits for a whole round of leaves to be complete: there is no
// workgroup barrier after the set-up.  A game's own sequence select -> evaluate -> expand/backup is what it
// is in the lock-step kernels and the network's outputs do not depend on which leaf shares a task or a head
// batch, so trees, visit counts and records are bit-identical to the two-kernel path.
// What bounds the kernel is the SIMDs' instruction issue: the matrix pipes are busy two thirds of the time, the
// other vector instructions (one per MFMA in a task, some 2 400 per search step) a third, and the two overlap
// little (an MFMA keeps the vector issue port for half of its sixteen cycles, and a SIMD serves its older wave
// first: tools/simd_share_probe.hip).  More waves, other priorities or more head servers therefore only move
// time between the games' queue waits and their service times; search steps and head batches run at raised
// priority because they are short and whole batches of games wait for them.
// Hand-offs: (p, v), slots, nodes and leaf positions are written, then a workgroup-scope release fence, then
// the queue entry (LDS atomics); the taker reads the entry, then an acquire fence.  A queue is a ring of
// FUSED_QUEUE words (game + 1, 0 = its writer has the slot but has not stored yet) with a head and a tail
// counter; a game has at most one entry in the three queues together, so the rings never wrap onto live entries.
// A worker that finds nothing to do for WATCHDOG_SPINS polls in a row during which no wave of the workgroup
// finished anything gives up (a fault elsewhere must not hang the GPU): it sets the abort word, every wave
// leaves, and bit 4 of the engine's error flags reports it.
// ---------------------------------------------------------------------------
#define FUSED_QUEUE 64                 // ring slots >= games per workgroup
#ifndef FUSED_WAVES
// worker waves per workgroup: 8 (two per SIMD) or 12 (three per SIMD: the kernel fits their 168 registers
// without scratch; measured 2 % slower, the SIMDs have no issue slots left for a third wave)
#define FUSED_WAVES 8
#endif
#define FUSED_THREADS (64 * FUSED_WAVES)
// Up to this many games per workgroup a leaf is evaluated alone and at once (measured at 200 simulations per move:
// +27 % positions/s at 4 games per workgroup, +20 % at 8-10, +12 % at 12, +6 % at 14, -1 % at 16)
#ifndef FUSED_FEW_GAMES
#define FUSED_FEW_GAMES 14
#endif
// up to this many games per workgroup (<= FUSED_FEW_GAMES) a worker runs the heads of its own board itself
// (measured: +6 % positions/s at 4-8 games per workgroup, +4 % at 10, -2.5 % at 12)
#ifndef FUSED_OWN_HEADS
#define FUSED_OWN_HEADS 10
#endif
#ifndef FUSED_HEADS_MIN
#define FUSED_HEADS_MIN 8              // boards waiting for their heads from which a worker prefers them to a search step
#endif
// the games' head planes live behind the network scratch: one slot per game
#define FX_PLANES S_BYTES
// per worker: the logits of the one board whose heads the worker runs itself (workgroups with few games), then
// that board's entry of the batch list
#define FX_OWN (FX_PLANES + 64 * PLANE_SLOT)
#define FX_OWN_BYTES 320
#ifdef AZ_EXP_TIMING
#define FX_TS (FX_OWN + FUSED_WAVES * FX_OWN_BYTES)     // tuning build: int [64], when each game entered the queue it is in
#define FX_END (FX_TS + 256)
#else
#define FX_END (FX_OWN + FUSED_WAVES * FX_OWN_BYTES)
#endif
#define FUSED_LDS_BYTES (OFF_BIAS + FX_END)
// the fp16-pair form (az_net_hx_dev.h): its own weights in front, the same hand-off layout behind them, then one
// strip of head planes per worker
#define FX_HX_STRIPS FX_END
#define FUSED_LDS_BYTES_HX (HX_W_BYTES + FX_HX_STRIPS + FUSED_WAVES * HX_STRIP_BYTES)
// the engine's 512-byte counter block (E.ring_count, uint32 words): 0 finished games in the ring, 1 error bits,
// 8..15 four 64-bit counters of this kernel (az_c4_move_fused_counters), 16..23 tuning builds, 64.. az_generic_select
#define FUSED_COUNTER_WORD 8
#define WATCHDOG_SPINS (1 << 20)       // x >= 128 cycles per poll: far beyond any wait of a healthy launch
enum { FC_LEAF_HEAD = 0, FC_LEAF_TAIL, FC_DONE, FC_ABORT, FC_READY_HEAD, FC_READY_TAIL, FC_SEARCHING, FC_GAMES,
       FC_PLANES_HEAD, FC_PLANES_TAIL, FC_HEADS_LOCK, FC_PROGRESS,
       FC_TOWERS /* tuning build: network tasks running on each SIMD, one byte per SIMD */,
       FC_BOARDS /* tuning build: boards this workgroup has put through the network */ };

__device__ __forceinline__ int lds_load(const int* a) {
    return __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(int* a, int x) {
    __hip_atomic_store(a, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_add(int* a, int x) {
    return __hip_atomic_fetch_add(a, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// one lane appends game gl to a ring (the caller has fenced its writes)
__device__ __forceinline__ void ring_push(int* ring, int* tail, int gl) {
    const int t = lds_add(tail, 1);
    lds_store(ring + (t & (FUSED_QUEUE - 1)), gl + 1);
}
// entry `at` of a ring whose head the caller has already moved past it; all lanes call it together
__device__ __forceinline__ int ring_take(int* ring, int at, int lane) {
    int* slot = ring + (at & (FUSED_QUEUE - 1));
    int x;
    while ((x = uni(lds_load(slot))) == 0) __builtin_amdgcn_s_sleep(1);      // its writer is one store behind
    if (lane == 0) lds_store(slot, 0);
    return x - 1;
}
// move a ring's head from h to h + n if nobody else has moved it; all lanes call it together
__device__ __forceinline__ bool ring_claim(int* head, int h, int n, int lane) {
    int won = 0;
    if (lane == 0) {
        int expect = h;
        won = __hip_atomic_compare_exchange_strong(head, &expect, h + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
    }
    return uni(won) != 0;
}

// A copy of the lane index that the compiler cannot see through: what a phase of the kernel computes from it (LDS
// addresses of weights, planes, queues) is computed inside that phase instead of being kept in registers across
// the other phases.  With it the kernel needs 177 registers instead of 233 (and fits 168 without scratch).
__device__ __forceinline__ int phase_lane(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}

#ifdef AZ_EXP_TIMING
// 0..2 cycles games waited in the leaf / plane / ready queue, 3..5 the number of waits; 6, 7 cycles and number of the
// two-board tasks that had their SIMD to themselves when they began and ended; 8, 9 the same for tasks that shared it
__device__ unsigned long long g_wait_dbg[10];
__device__ unsigned long long g_wg_dbg[2 * 1024];     // per workgroup of the last launch: cycles until its last game was done, boards evaluated
__device__ unsigned long long g_done_dbg[8];     // cycles from the start until 1 / 50 % / 75 % / 90 % / all games of a workgroup are done
#endif
#ifndef FUSED_MIN_WAVES       // waves per SIMD the register budget is cut for; diagnostic builds raise it to make the
#define FUSED_MIN_WAVES (FUSED_WAVES / 4)     // kernel spill (3: 168 registers, 4: 128)
#endif
// HX: the network at fp32 grade (operands as fp16 pairs, az_net_hx_dev.h; blob = pack_c4_blob_hx): every leaf is a
// one-board task whose worker also runs its heads -- the policy of a workgroup with few games, for any number of
// games.  Everything else (queues, search steps, hand-offs) is the same code.
template <bool HX>
__global__ void __launch_bounds__(FUSED_THREADS, FUSED_MIN_WAVES)
k_c4_move_fused(Dev E, const uint8_t* blob, int num_blocks, uint64_t* leaf_bb, int32_t* leaf_info, float* p,
                float* v, int num_sims, int gpw) {
    constexpr int GS = 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* wlds = lds;
    uint8_t* scr = lds + (HX ? HX_W_BYTES : OFF_BIAS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    {
        uint4* z = reinterpret_cast<uint4*>(scr);
        for (int i = tid; i < FX_END / 16; i += FUSED_THREADS) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if constexpr (HX) {
        if (tid < 100) reinterpret_cast<float*>(scr + S_BIAS)[tid] = reinterpret_cast<const float*>(blob + HX_OFF_BIAS)[tid];
        hx_load_weights(blob, wlds, tid, FUSED_THREADS);       // once per move
    } else {
        if (tid < 99) reinterpret_cast<float*>(scr + S_BIAS)[tid] = reinterpret_cast<const float*>(blob + OFF_BIAS)[tid];
        if (tid == 99) reinterpret_cast<float*>(scr + S_BIAS)[99] = reinterpret_cast<const float*>(blob + OFF_B2)[0];
        if (tid < 256) {
            reinterpret_cast<float*>(scr + S_B1)[tid] = reinterpret_cast<const float*>(blob + OFF_B1)[tid];
            reinterpret_cast<float*>(scr + S_W2)[tid] = reinterpret_cast<const float*>(blob + OFF_W2)[tid];
        }
        if (tid < 64) reinterpret_cast<float*>(scr + S_BP)[tid] = reinterpret_cast<const float*>(blob + OFF_BP)[tid];
        if (tid < NTHREADS) load_weights(blob, wlds, tid);         // once per move
    }

    // LDS words of the hand-offs (all zero now): the task-list area holds the games' select counts, the value
    // partial-sum area the two rings
    int* nsel = reinterpret_cast<int*>(scr + S_TL);                      // [64] selects done
    int* gmap = nsel + 64;                                               // [64] the workgroup's games (E.order)
    int* leafq = reinterpret_cast<int*>(scr + S_LG + 32 * 64 * 4);       // [FUSED_QUEUE]
    int* readyq = leafq + FUSED_QUEUE;                                   // [FUSED_QUEUE]
    int* planeq = readyq + FUSED_QUEUE;                                  // [FUSED_QUEUE] boards waiting for their heads
    int* batch = planeq + FUSED_QUEUE;                                   // [32] the boards of the head batch in flight
    uint8_t* planes = scr + FX_PLANES;                                   // [64][PLANE_SLOT]
    int* ctr = reinterpret_cast<int*>(scr + S_WT);                       // FC_* words
    static_assert(8 * MAXIT * 2 >= 128 && (3 * FUSED_QUEUE + 32) * 4 <= 8 * 32 * 4, "LDS areas reused by the hand-off words");
    uint64_t* lbb = reinterpret_cast<uint64_t*>(scr + S_LBB);            // [64][2] leaf positions
    const int wave_u = uni(wave);          // (a scalar)
    const int wave_id = (int)blockIdx.x * FUSED_WAVES + wave_u;
    if (wave == 0) {
        // the workgroup's games: the leading non-negative entries of its row of E.order; all start ready for
        // their first select
        const int g = lane < gpw ? E.order[(size_t)blockIdx.x * gpw + lane] : -1;
        gmap[lane] = g;
        if (g >= 0) readyq[lane] = lane + 1;
        const int n = __popcll(__ballot(g >= 0));
        if (lane == 0) { ctr[FC_READY_TAIL] = n; ctr[FC_GAMES] = n; }
    }
    __syncthreads();                       // the only time the waves meet
    const int n_valid = ctr[FC_GAMES];

    const float* sbias = reinterpret_cast<const float*>(scr + S_BIAS);
#ifdef AZ_EXP_TIMING
    const long long t_start = clock64();
#endif
    // one search step of game gl by this lane's group of eight
#ifdef AZ_EXP_TIMING
    int* tstamp = reinterpret_cast<int*>(scr + FX_TS);
    if (wave == 0) tstamp[lane] = (int)clock64();
    long long w_leaf = 0, w_plane = 0, w_ready = 0, n_leaf = 0, n_plane = 0, n_ready = 0;
    const int simd_shift = 8 * (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));      // HW_ID bits 5:4
#endif
    auto step = [&](int lane, int gl, auto tag_expand, bool sel) {
        const int sub = lane & (GS - 1);
        if (!AZ_IDX_OK(E, gl >= 0 && gl < 64, 7, gl)) return;
        const int g = gmap[gl];
        if (!AZ_IDX_OK(E, g >= 0 && g < E.G, 8, g)) return;
        Slot* s = E.slots + g;
        uint4* pool = E.pool + (size_t)g * E.cap * (GameC4::A + 1);
        const int kind = sim_step_game<GameC4, GS, decltype(tag_expand)::value, true, true>(
            E, p, v, leaf_bb, leaf_info, g, lane, sub, s, pool, s->path, wave_id, lbb + 2 * gl, sel);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (sub == 0) {
            if (!sel || num_sims <= 0) {                       // (no simulations asked for: the first select is all)
#ifdef AZ_EXP_TIMING
                {
                    const int d = lds_add(ctr + FC_DONE, 1) + 1, nv = ctr[FC_GAMES];
                    const long long now = clock64() - t_start;
                    if (d == (nv + 1) / 2) atomicAdd(&g_done_dbg[0], (unsigned long long)now);
                    if (d == (3 * nv + 3) / 4) atomicAdd(&g_done_dbg[1], (unsigned long long)now);
                    if (d == (9 * nv + 9) / 10) atomicAdd(&g_done_dbg[2], (unsigned long long)now);
                    if (d == nv) { atomicAdd(&g_done_dbg[3], (unsigned long long)now); atomicAdd(&g_done_dbg[4], 1ull); }
                    if (d == 1) atomicAdd(&g_done_dbg[5], (unsigned long long)now);
                }
#else
                lds_add(ctr + FC_DONE, 1);
#endif
            } else {
                nsel[gl] += 1;
#ifdef AZ_EXP_TIMING
                tstamp[gl] = (int)clock64();
#endif
                if (kind == AZ_LEAF_EVAL || kind == AZ_LEAF_ROOT) ring_push(leafq, ctr + FC_LEAF_TAIL, gl);
                else ring_push(readyq, ctr + FC_READY_TAIL, gl);     // terminal (backed up already) or cached
            }
        }
    };

#ifdef AZ_EXP_TIMING     // tuning aid (tools/exp_fused_timing.py): what the waves spend their cycles on
    long long t_task = 0, t_search = 0, t_idle = 0, t_heads = 0, t_fence = 0, t0 = clock64(), t1;
    unsigned long long n_pair = 0, n_single = 0, n_step = 0, n_stepped = 0, n_hbatch = 0;
#define TICK(acc) do { t1 = clock64(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define TICK(acc) do { } while (0)
#endif
    const bool few_games = HX || gpw <= FUSED_FEW_GAMES;
    // what this worker did (wave-uniform counts, added to the engine's counters when it leaves): two-board tasks,
    // one-board tasks, head batches, boards in head batches
    uint32_t did_pair = 0, did_single = 0, did_hbatch = 0, did_hboards = 0;
    for (int spins = 0, seen_progress = 0;;) {
        if (uni(lds_load(ctr + FC_ABORT)) || uni(lds_load(ctr + FC_DONE)) >= n_valid) break;
        const int lh = uni(lds_load(ctr + FC_LEAF_HEAD)), leaves = uni(lds_load(ctr + FC_LEAF_TAIL)) - lh;
        const int rh = uni(lds_load(ctr + FC_READY_HEAD)), ready = uni(lds_load(ctr + FC_READY_TAIL)) - rh;
        const int ph = uni(lds_load(ctr + FC_PLANES_HEAD)), waiting = uni(lds_load(ctr + FC_PLANES_TAIL)) - ph;
        // a lone leaf is taken only when no second one can turn up soon: nothing ready, nothing being searched or
        // waiting for its heads
        // (a workgroup with few games has idle workers and idle matrix pipes: there a leaf is taken at once and alone,
        // two one-board tasks side by side end sooner than one two-board task)
        if (leaves >= 2 || (leaves == 1 && (few_games || (ready <= 0 && waiting <= 0 && uni(lds_load(ctr + FC_SEARCHING)) == 0)))) {
            const int take = (leaves >= 2 && !few_games) ? 2 : 1;
            if (!ring_claim(ctr + FC_LEAF_HEAD, lh, take, lane)) continue;
            TICK(t_idle);
            const int g0 = ring_take(leafq, lh, lane), g1 = take == 2 ? ring_take(leafq, lh + 1, lane) : -1;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef AZ_EXP_TIMING
            if (lane == 0) {
                const int now = (int)clock64();
                w_leaf += (unsigned)(now - tstamp[g0]); ++n_leaf;
                if (g1 >= 0) { w_leaf += (unsigned)(now - tstamp[g1]); ++n_leaf; }
            }
            const int towers_before = (uni(lane == 0 ? lds_add(ctr + FC_TOWERS, 1 << simd_shift) : 0) >> simd_shift) & 255;
            if (lane == 0) lds_add(ctr + FC_BOARDS, take);
            const long long task_t0 = clock64();
#endif
            if (!AZ_IDX_OK(E, g0 >= 0 && g0 < 64 && g1 >= -1 && g1 < 64, 9, (g0 << 8) | (g1 & 255))) break;
            __bf16* hp0 = reinterpret_cast<__bf16*>(planes + g0 * PLANE_SLOT);
            const int tl = phase_lane(lane);
            if constexpr (HX) {
                float* strip = reinterpret_cast<float*>(scr + FX_HX_STRIPS + wave_u * HX_STRIP_BYTES);
                hx_trunk_task(wlds, sbias, num_blocks, lbb + 2 * g0, strip, tl);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");          // the strip, before this wave reads it back
                const int g = gmap[g0];
                __builtin_amdgcn_s_setprio(3);
                hx_heads_own(blob, sbias, strip, lbb + 2 * g0, p + (size_t)g * 7, v + g, tl);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // (p, v), before the game is queued
#ifdef AZ_EXP_TIMING
                if (tl == 0) tstamp[g0] = (int)clock64();
#endif
                if (tl == 0) ring_push(readyq, ctr + FC_READY_TAIL, g0);
                ++did_hbatch; ++did_hboards;
            } else {
            if (g1 >= 0) {
                const int g_mine = (tl & 8) ? g1 : g0;
                __bf16* hp1 = reinterpret_cast<__bf16*>(planes + g1 * PLANE_SLOT);
                trunk_task<ROWS>(wlds, sbias, num_blocks, lbb + 2 * g_mine, hp0, hp0 + 96, hp1, hp1 + 96, tl);
            } else {
                trunk_task<ROWS / 2>(wlds, sbias, num_blocks, lbb + 2 * g0, hp0, hp0 + 96, nullptr, nullptr, tl);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the planes, before the boards are queued
#ifdef AZ_EXP_TIMING
            {
                const int towers_after = ((uni(lane == 0 ? lds_add(ctr + FC_TOWERS, -(1 << simd_shift)) : 0) >> simd_shift) & 255) - 1;
                const unsigned long long dt = (unsigned long long)(clock64() - task_t0);
                if (lane == 0 && g1 >= 0 && towers_before == 0 && towers_after == 0) { atomicAdd(&g_wait_dbg[6], dt); atomicAdd(&g_wait_dbg[7], 1ull); }
                if (lane == 0 && g1 >= 0 && towers_before >= 1 && towers_after >= 1) { atomicAdd(&g_wait_dbg[8], dt); atomicAdd(&g_wait_dbg[9], 1ull); }
                if (lane == 0) { tstamp[g0] = (int)clock64(); if (g1 >= 0) tstamp[g1] = tstamp[g0]; }
            }
#endif
            if (gpw <= FUSED_OWN_HEADS) {
                // very few games: nobody to share a head batch with, so the worker runs its board's heads itself (its own
                // logits area, no lock, no queue) and the game goes straight back to the ready queue
                float* own_lg = reinterpret_cast<float*>(scr + FX_OWN + wave_u * FX_OWN_BYTES);
                int* own_batch = reinterpret_cast<int*>(own_lg + 64);
                if (tl == 0) own_batch[0] = g0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_s_setprio(3);
                heads_batch(wlds, scr, own_lg, planes, own_batch, 1, gmap, lbb, p, v, tl);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (p, v), before the game is queued
#ifdef AZ_EXP_TIMING
                if (tl == 0) tstamp[g0] = (int)clock64();
#endif
                if (tl == 0) ring_push(readyq, ctr + FC_READY_TAIL, g0);
                ++did_hbatch; ++did_hboards;
            } else if (lane == 0) {
                ring_push(planeq, ctr + FC_PLANES_TAIL, g0);
                if (g1 >= 0) ring_push(planeq, ctr + FC_PLANES_TAIL, g1);
            }
            }
            TICK(t_task);
            if (g1 >= 0) ++did_pair; else ++did_single;
#ifdef AZ_EXP_TIMING
            if (g1 >= 0) ++n_pair; else ++n_single;
#endif
            if (lane == 0) lds_add(ctr + FC_PROGRESS, 1);
            spins = 0;
            continue;
        }
        // The fully connected heads of all boards that wait for them, at most 32, by one wave at a time: a batch costs
        // what two boards cost.  Preferred to a search step once FUSED_HEADS_MIN boards wait, else taken when there
        // is nothing to search.
        if ((waiting >= FUSED_HEADS_MIN || (waiting > 0 && ready <= 0)) && uni(lds_load(ctr + FC_HEADS_LOCK)) == 0 &&
            ring_claim(ctr + FC_HEADS_LOCK, 0, 1, lane)) {
            // (only the holder of the lock moves this ring's head)
            const int ph2 = uni(lds_load(ctr + FC_PLANES_HEAD));
            int n = uni(lds_load(ctr + FC_PLANES_TAIL)) - ph2;
            n = n < 32 ? n : 32;
            TICK(t_idle);
            if (n > 0) {
                const int hl = phase_lane(lane);
                if (hl == 0) { lds_store(ctr + FC_PLANES_HEAD, ph2 + n); lds_add(ctr + FC_SEARCHING, n); }
                if (hl < n) {
                    int* slot = planeq + ((ph2 + hl) & (FUSED_QUEUE - 1));
                    int x;
                    while ((x = lds_load(slot)) == 0) __builtin_amdgcn_s_sleep(1);      // its writer is one store behind
                    lds_store(slot, 0);
                    if (!AZ_IDX_OK(E, x >= 1 && x <= 64, 10, x)) x = 1;
                    batch[hl] = x - 1;
#ifdef AZ_EXP_TIMING
                    w_plane += (unsigned)((int)clock64() - tstamp[x - 1]); ++n_plane;
#endif
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_s_setprio(3);     // a short chain of dependent steps that up to 32 games wait for
                heads_batch(wlds, scr, reinterpret_cast<float*>(scr + S_LG), planes, batch, n, gmap, lbb, p, v, hl);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (p, v), before the games are queued
#ifdef AZ_EXP_TIMING
                if (hl < n) tstamp[batch[hl]] = (int)clock64();
#endif
                if (hl < n) ring_push(readyq, ctr + FC_READY_TAIL, batch[hl]);
                if (hl == 0) lds_add(ctr + FC_SEARCHING, -n);
                ++did_hbatch; did_hboards += (uint32_t)n;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) lds_store(ctr + FC_HEADS_LOCK, 0);
            TICK(t_heads);
#ifdef AZ_EXP_TIMING
            ++n_hbatch;
#endif
            if (lane == 0) lds_add(ctr + FC_PROGRESS, 1);
            spins = 0;
            continue;
        }
        if (ready > 0) {
            const int take = ready < 8 ? ready : 8;
            if (!ring_claim(ctr + FC_READY_HEAD, rh, take, lane)) continue;
            TICK(t_idle);
            if (lane == 0) lds_add(ctr + FC_SEARCHING, take);
            __builtin_amdgcn_s_setprio(3);     // few instructions, all on the critical path of some network task
            // lane i < take fetches entry i, then group i of eight lanes takes it over
            int mine = 0;
            if (lane < take) {
                int* slot = readyq + ((rh + lane) & (FUSED_QUEUE - 1));
                while ((mine = lds_load(slot)) == 0) __builtin_amdgcn_s_sleep(1);      // its writer is one store behind
                lds_store(slot, 0);
#ifdef AZ_EXP_TIMING
                w_ready += (unsigned)((int)clock64() - tstamp[mine - 1]); ++n_ready;
#endif
            }
            const int gl = __shfl(mine, lane >> 3) - 1;                              // -1: no game for this group
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (gl >= 0) {
                const int done = nsel[gl];
                const int sl = phase_lane(lane);
                if (done == 0) step(sl, gl, std::false_type{}, true);          // the game's first select
                else step(sl, gl, std::true_type{}, done < num_sims);
            }
            __builtin_amdgcn_s_setprio(0);
            if (lane == 0) lds_add(ctr + FC_SEARCHING, -take);             // after the pushes of this step
            TICK(t_search);
#ifdef AZ_EXP_TIMING
            ++n_step; n_stepped += take;
#endif
            if (lane == 0) lds_add(ctr + FC_PROGRESS, 1);
            spins = 0;
            continue;
        }
        __builtin_amdgcn_s_sleep(2);
        {
            // idle: the watchdog counts polls during which NO wave of the workgroup finished anything (a wave may
            // well have nothing to do for a whole launch: a workgroup with one game keeps one or two waves busy)
            const int progress = uni(lds_load(ctr + FC_PROGRESS));
            if (progress != seen_progress) { seen_progress = progress; spins = 0; }
        }
        if (++spins > WATCHDOG_SPINS) {
            if (lane == 0) {
                lds_store(ctr + FC_ABORT, 1);
                atomicOr(reinterpret_cast<unsigned int*>(E.ring_count + 1), 16u);
            }
            break;
        }
    }
    if (lane == 0) {
        unsigned long long* fc = reinterpret_cast<unsigned long long*>(E.ring_count + FUSED_COUNTER_WORD);
        if (did_pair) atomicAdd(fc + 0, (unsigned long long)did_pair);
        if (did_single) atomicAdd(fc + 1, (unsigned long long)did_single);
        if (did_hbatch) { atomicAdd(fc + 2, (unsigned long long)did_hbatch); atomicAdd(fc + 3, (unsigned long long)did_hboards); }
    }
#ifdef AZ_EXP_TIMING
    TICK(t_idle);
    if (tid == 0 && blockIdx.x < 1024) {
        g_wg_dbg[2 * blockIdx.x] = (unsigned long long)(clock64() - t_start);
        g_wg_dbg[2 * blockIdx.x + 1] = (unsigned long long)lds_load(ctr + FC_BOARDS);
    }
    atomicAdd(&g_wait_dbg[0], (unsigned long long)w_leaf); atomicAdd(&g_wait_dbg[1], (unsigned long long)w_plane);
    atomicAdd(&g_wait_dbg[2], (unsigned long long)w_ready); atomicAdd(&g_wait_dbg[3], (unsigned long long)n_leaf);
    atomicAdd(&g_wait_dbg[4], (unsigned long long)n_plane); atomicAdd(&g_wait_dbg[5], (unsigned long long)n_ready);
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(E.ring_count + 16);
        atomicAdd(dbg + 0, (unsigned long long)t_task);
        atomicAdd(dbg + 1, (unsigned long long)t_search);
        atomicAdd(dbg + 2, (unsigned long long)t_idle);
        atomicAdd(dbg + 3, 1ull);
        atomicAdd(&g_net_dbg[0], n_pair);
        atomicAdd(&g_net_dbg[1], n_single);
        atomicAdd(&g_net_dbg[2], n_step);
        atomicAdd(&g_net_dbg[3], n_stepped);
        atomicMax(&g_net_dbg[4], (unsigned long long)(t_task + t_heads + t_fence + t_search + t_idle));
        atomicAdd(&g_net_dbg[6], (unsigned long long)t_fence);
        atomicAdd(&g_net_dbg[7], n_hbatch);
        atomicAdd(&g_net_dbg[5], (unsigned long long)t_heads);
    }
#endif
#undef TICK
}

// num_sims whole simulations per game in one launch, UniformModel evaluated in place
template <class GM, int GS>
__global__ void __launch_bounds__(BLOCK_THREADS) k_run_uniform_p(Dev E, int num_sims) {
    GROUP_PROLOGUE
    uint32_t alloc = s->alloc, err = s->err;
    const uint32_t root = s->root;
    const uint64_t nx0 = s->nx, pv0 = s->pv;
    const int nm0 = s->nm;
    uint32_t sims = 0, evals = 0, steps = 0, root_ns = s->root_ns;
    if (root == AZ_NO_NODE) return;
    for (int it = 0; it < num_sims; ++it) {
        uint64_t nx = nx0, pv = pv0;
        int nm = nm0, depth, kind;
        float vterm;
        uint4 cq_unused;
        select_leaf_g<GM, GS>(E, pool, root, nx, pv, nm, path, sub, kind, depth, vterm, steps, root_ns, false, cq_unused);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (kind == AZ_LEAF_EVAL) {
            const uint32_t idx = expand_node_g<GM, GS>(E, pool, alloc, nx, pv, nullptr, err, sub);
            const uint32_t last = path[depth - 1];
            if (idx != AZ_NO_NODE && sub == 0)
                reinterpret_cast<uint32_t*>(&pool[(size_t)(last >> 4) * (GM::A + 1) + (last & 15u)])[3] = idx;
            backup_g<GM, GS>(E, pool, depth, path, -0.0f, sub);           // v = 0, then v *= -1
            evals += 1;
        } else if (kind == AZ_LEAF_TERMINAL) {
            backup_g<GM, GS>(E, pool, depth, path, vterm, sub);
        }
        sims += 1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");        // own stores before next select's loads
    }
    if (sub == 0) {
        s->alloc = alloc; s->err = err; s->leaf_kind = AZ_LEAF_NONE; s->root_ns = root_ns;
        s->move_sims += sims; s->st_sims += sims; s->st_evals += evals; s->st_steps += steps;
    }
}

// ---------------------------------------------------------------------------
// User-defined games (AZ_GAME_GENERIC): the device keeps the tree and does select / expand / backup with the
// same numerics; the rules stay with the caller's Game object on the host.  A node has 15 edge records + the
// count record = 256 B.  The walk needs no rules: the reference never expands a terminal node
// (mcts.py:131-134), so descending until an edge has no child node ends exactly where mcts.py:123-127 ends;
// the host replays the returned action indices on a clone of its game and decides terminal / evaluate.
// One wavefront per call, one game per call: this serves the single-game run_mcts mirror, not throughput.
// ---------------------------------------------------------------------------
#define GEN_A 15
#define GEN_AS 16
struct GenPrior { float p[GEN_A]; };
struct GenNoise { double eta[GEN_A]; };

// mcts.py:114-117: expand the root when it has no node yet (n_edges > 0), then blend the noise into its priors
__global__ void __launch_bounds__(64) k_generic_root(Dev E, int g, int n_edges, GenPrior pr, GenNoise no, int use_noise) {
    const int lane = lane_id();
    Slot* s = E.slots + g;
    uint4* pool = E.pool + (size_t)g * E.cap * GEN_AS;
    uint32_t root = s->root, alloc = s->alloc, err = s->err;
    if (root == AZ_NO_NODE && n_edges > 0) {
        if (alloc < (uint32_t)E.cap) {
            root = alloc++;
            if (lane < GEN_AS) {
                uint4 rec = make_uint4(0u, 0u, lane < n_edges ? __float_as_uint(pr.p[lane < GEN_A ? lane : 0]) : 0u, AZ_NO_NODE);
                if (lane == GEN_AS - 1) rec = make_uint4((uint32_t)n_edges, 0u, 0u, 0u);
                pool[(size_t)root * GEN_AS + lane] = rec;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        } else err |= 1u;
    }
    if (root != AZ_NO_NODE && use_noise) {
        const int ne = (int)pool[(size_t)root * GEN_AS + GEN_AS - 1].x;
        if (lane < ne) {                                   // mcts.py:88-91, as root_noise
            uint32_t* pp = reinterpret_cast<uint32_t*>(&pool[(size_t)root * GEN_AS + lane]) + 2;
            const float a = E.one_minus_eps * __uint_as_float(*pp);
            *pp = __float_as_uint((float)((double)a + E.eps * no.eta[lane]));
        }
    }
    if (lane == 0) { s->root = root; s->alloc = alloc; s->err = err; s->leaf_kind = AZ_LEAF_NONE; s->move_sims = 0; }
}

// mcts.py:119-127 without the rules: out[0] = depth, out[1 + d] = action index chosen at level d
__global__ void __launch_bounds__(64) k_generic_select(Dev E, int g, uint32_t* out) {
    const int lane = lane_id();
    Slot* s = E.slots + g;
    const uint4* pool = E.pool + (size_t)g * E.cap * GEN_AS;
    uint32_t node = s->root;
    int depth = 0;
    while (node != AZ_NO_NODE && depth < PATH_LEN) {
        uint4 e = make_uint4(0u, 0u, 0u, AZ_NO_NODE);
        if (lane < GEN_AS) e = pool[(size_t)node * GEN_AS + lane];
        const int n_edges = (int)rdlane(e.x, GEN_AS - 1);
        const uint32_t Ns = rdlane(e.y, GEN_AS - 1);
        double cs, sq;
        lut_lookup(E, Ns, cs, sq);
        const bool act = lane < n_edges;
        double score = -INFINITY;
        if (act) score = puct_score(e.x, __uint_as_float(e.y), __uint_as_float(e.z), cs, sq);
        const double mx = row16_max(score);
        const unsigned long long hit = __ballot(act && score == mx) & 0xFFFFull;
        const int slot = __ffsll((long long)hit) - 1;                   // np.argmax: first maximum
        if (lane == 0) { s->path[depth] = (node << 4) | (uint32_t)slot; out[1 + depth] = (uint32_t)slot; }
        depth += 1;
        node = rdlane(e.w, slot);
    }
    if (node != AZ_NO_NODE) {
        // the path stack is full and the walk is still inside the tree: nothing may be expanded or backed up from
        // here (it would replace an existing child and lose its subtree).  Error bit 5; the host call fails.
        if (lane == 0) { s->leaf_depth = 0; s->leaf_kind = AZ_LEAF_NONE; out[0] = 0xFFFFFFFFu; atomicOr(E.ring_count + 1, 32u); }
        return;
    }
    if (lane == 0) { s->leaf_depth = depth; s->leaf_kind = AZ_LEAF_EVAL; out[0] = (uint32_t)depth; s->st_steps += (uint64_t)depth; }
}

// mcts.py:131-158: n_edges > 0: the leaf gets a node with the host's priors (mcts.py:138); v_edge is the value the
// leaf's edge receives (terminal value, or the network's value already negated, mcts.py:134,144)
__global__ void __launch_bounds__(64) k_generic_expand_backup(Dev E, int g, int n_edges, GenPrior pr, float v_edge) {
    const int lane = lane_id();
    Slot* s = E.slots + g;
    uint4* pool = E.pool + (size_t)g * E.cap * GEN_AS;
    const int depth = s->leaf_depth;
    uint32_t alloc = s->alloc, err = s->err;
    if (s->leaf_kind != AZ_LEAF_EVAL || depth <= 0) return;
    if (n_edges > 0) {
        if (alloc < (uint32_t)E.cap) {
            const uint32_t idx = alloc++;
            if (lane < GEN_AS) {
                uint4 rec = make_uint4(0u, 0u, lane < n_edges ? __float_as_uint(pr.p[lane < GEN_A ? lane : 0]) : 0u, AZ_NO_NODE);
                if (lane == GEN_AS - 1) rec = make_uint4((uint32_t)n_edges, 0u, 0u, 0u);
                pool[(size_t)idx * GEN_AS + lane] = rec;
            }
            const uint32_t last = s->path[depth - 1];
            if (lane == 0) reinterpret_cast<uint32_t*>(&pool[(size_t)(last >> 4) * GEN_AS + (last & 15u)])[3] = idx;
        } else err |= 1u;
    }
    for (int d = lane; d < depth; d += 64) {                            // levels are distinct nodes
        const uint32_t ent = s->path[d];
        const uint32_t nd = ent >> 4, sl = ent & 15u;
        uint2* ep = reinterpret_cast<uint2*>(&pool[(size_t)nd * GEN_AS + sl]);
        uint32_t* ns = reinterpret_cast<uint32_t*>(&pool[(size_t)nd * GEN_AS + GEN_AS - 1]) + 1;
        uint2 nw = *ep;
        const float v = ((depth - 1 - d) & 1) ? -v_edge : v_edge;
        nw.x += 1u;
        nw.y = __float_as_uint(__uint_as_float(nw.y) + v);
        *ep = nw;
        *ns += 1u;
    }
    if (lane == 0) {
        s->alloc = alloc; s->err = err; s->leaf_kind = AZ_LEAF_NONE;
        s->st_sims += 1; s->move_sims += 1;
        if (n_edges > 0) s->st_evals += 1;
    }
}

// ---------------------------------------------------------------------------
// stateless bulk kernels (one thread per position)
// ---------------------------------------------------------------------------
template <class GM>
__global__ void k_from_history(const uint8_t* hist, const int32_t* len, int stride, int n, uint64_t* bb,
                               int32_t* num_moves) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t nx = 0, pv = 0;
    int nm = 0;
    const int L = len[i];
    for (int k = 0; k < L; ++k) GM::play(nx, pv, nm, hist[(size_t)i * stride + k]);
    bb[2 * (size_t)i] = nx; bb[2 * (size_t)i + 1] = pv;
    num_moves[i] = nm;
}

template <class GM>
__global__ void k_query(const uint64_t* bb, const int32_t* num_moves, int n, int32_t* legal, int32_t* term,
                        int32_t* winner) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t nx = bb[2 * (size_t)i], pv = bb[2 * (size_t)i + 1];
    const int nm = num_moves[i];
    const int w = GM::winner(nx, pv, nm);
    if (legal) legal[i] = (int32_t)GM::legal(nx, pv);
    if (term) term[i] = w != AZ_W_NONE;
    if (winner) winner[i] = w;
}

template <class GM>
__global__ void k_step(uint64_t* bb, int32_t* num_moves, const int32_t* action, int n, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t nx = bb[2 * (size_t)i], pv = bb[2 * (size_t)i + 1];
    int nm = num_moves[i];
    const int a = action[i];
    const bool ok = a >= 0 && a < GM::A && ((GM::legal(nx, pv) >> a) & 1u);   // game.py:31
    if (ok) {
        GM::play(nx, pv, nm, a);
        bb[2 * (size_t)i] = nx; bb[2 * (size_t)i + 1] = pv;
        num_moves[i] = nm;
    }
    if (status) status[i] = ok ? 0 : -1;
}

template <class GM>
__global__ void k_render(const uint64_t* bb, int n, float* x, uint8_t* valid) {
    constexpr int CELLS = GM::H * GM::W;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / (2 * CELLS), r = t % (2 * CELLS);
    if (i >= n) return;
    const uint64_t b = bb[2 * (size_t)i + r / CELLS];
    const int c = r % CELLS;
    if (x) x[t] = GM::cell(b, c / GM::W, c % GM::W) ? 1.0f : 0.0f;
    if (valid && r < GM::A)
        valid[(size_t)i * GM::A + r] = (GM::legal(bb[2 * (size_t)i], bb[2 * (size_t)i + 1]) >> r) & 1u;
}

// replay_buffer.py:113-147 for every state of a batch of finished games: thread = state
template <class GM>
__global__ void k_examples(const az_game_record* __restrict__ recs, int nrec, const int32_t* __restrict__ state_offset,
                           int total, float* __restrict__ x, float* __restrict__ p, float* __restrict__ z,
                           uint8_t* __restrict__ valid, int32_t* __restrict__ version) {
    constexpr int CELLS = GM::H * GM::W;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total) return;
    int lo = 0, hi = nrec;                           // last record whose first state is <= s
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (state_offset[mid] <= s) lo = mid; else hi = mid;
    }
    const az_game_record& r = recs[lo];
    const int m = s - state_offset[lo];
    uint64_t nx = 0, pv = 0;
    int nm = 0;
    for (int k = 0; k < m; ++k) GM::play(nx, pv, nm, r.history[k]);     // Game(history[:move_index])
    const uint32_t legal = GM::legal(nx, pv);
    // p[valid_actions] = n / np.sum(n), n as stored: int16 (replay_buffer.py:30-35), quotient in fp64
    const int ne = r.num_edges[m];
    long sum = 0;
    for (int i = 0; i < ne; ++i) sum += (int16_t)r.visits[m][i];
    for (int a = 0; a < GM::A; ++a) {
        p[(size_t)s * GM::A + a] = 0.0f;
        valid[(size_t)s * GM::A + a] = (legal >> a) & 1u;
    }
    for (int i = 0; i < ne; ++i)
        p[(size_t)s * GM::A + az_nth_set_bit(legal, i)] = (float)((double)(int16_t)r.visits[m][i] / (double)sum);
    // outcome for the player to move (game.py:42-47)
    z[s] = r.winner < 0 ? 0.0f : ((m & 1) == r.winner ? 1.0f : -1.0f);
    if (version) version[s] = r.model_version;
    for (int c = 0; c < CELLS; ++c) {
        x[(size_t)s * 2 * CELLS + c] = GM::cell(nx, c / GM::W, c % GM::W) ? 1.0f : 0.0f;
        x[(size_t)s * 2 * CELLS + CELLS + c] = GM::cell(pv, c / GM::W, c % GM::W) ? 1.0f : 0.0f;
    }
}

template <int A>
__global__ void k_masked_softmax(const float* logits, const int32_t* info, int n, float* p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t legal = ((uint32_t)info[i] >> 16) & 0x1FFu;
    float m = -INFINITY;
    for (int a = 0; a < A; ++a) if ((legal >> a) & 1u) m = fmaxf(m, logits[(size_t)i * A + a]);
    float e[A], sum = 0.f;
    for (int a = 0; a < A; ++a) {
        e[a] = ((legal >> a) & 1u) ? expf(logits[(size_t)i * A + a] - m) : 0.f;
        sum += e[a];
    }
    for (int a = 0; a < A; ++a) p[(size_t)i * A + a] = ((legal >> a) & 1u) ? e[a] / sum : 0.f;
}

template <int A>
__global__ void k_ucb(Dev E, const uint32_t* N, const float* W, const float* P, const int32_t* ne, int n,
                      double* score, int32_t* argmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t Ns = 0;
    const int k = ne[i];
    for (int a = 0; a < k; ++a) Ns += N[(size_t)i * A + a];
    double cs, sq;
    lut_lookup(E, Ns, cs, sq);
    int best = 0;
    double bs = 0;
    for (int a = 0; a < k; ++a) {
        const double sc = puct_score(N[(size_t)i * A + a], W[(size_t)i * A + a], P[(size_t)i * A + a], cs, sq);
        score[(size_t)i * A + a] = sc;
        if (a == 0 || sc > bs) { bs = sc; best = a; }
    }
    for (int a = k; a < A; ++a) score[(size_t)i * A + a] = 0.0;
    argmax[i] = best;
}

// finished-game records from the ring to a caller's device buffer, 16 bytes per thread; head_out = {records
// copied, error bits}.  The ring counter is reset by the host code after this kernel, in stream order.
__global__ void __launch_bounds__(256)
k_take_finished(const az_game_record* __restrict__ ring, const uint32_t* __restrict__ ring_count, uint32_t ring_cap,
                az_game_record* __restrict__ out, uint32_t max_records, int32_t* __restrict__ head_out) {
    static_assert(sizeof(az_game_record) % 16 == 0, "records are copied as 16-byte words");
    constexpr uint32_t WORDS = sizeof(az_game_record) / 16;
    uint32_t count = ring_count[0], flags = ring_count[1];
    if (count > ring_cap) count = ring_cap;
    if (count > max_records) { count = max_records; flags |= 2u; }       // the rest is dropped: report it
    const uint4* src = reinterpret_cast<const uint4*>(ring);
    uint4* dst = reinterpret_cast<uint4*>(out);
    const size_t total = (size_t)count * WORDS;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { head_out[0] = (int32_t)count; head_out[1] = (int32_t)flags; }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline dim3 wave_grid(int G) { return dim3((G + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }
// lanes per game of the packed search kernels: the smallest power of two that holds a node
// (A+1 records) issues the fewest vector instructions; larger groups put more wavefronts on each
// SIMD to hide memory latency.  AZ_GROUP_SIZE (8/16/32/64) overrides the default for tuning.
template <class GM> struct GroupSize { static constexpr int min_value = GM::A + 1 <= 8 ? 8 : 16; };
static int group_size_for(int min_gs, int G) {
    static int env = -1;
    if (env < 0) { const char* v = getenv("AZ_GROUP_SIZE"); env = v ? atoi(v) : 0; }
    (void)G;
    int gs = env ? env : min_gs;
    if (gs != 8 && gs != 16 && gs != 32 && gs != 64) gs = 16;
    return gs < min_gs ? min_gs : gs;
}
static inline dim3 group_grid(int G, int gs) {
    const int per_block = WAVES_PER_BLOCK * (64 / gs);
    return dim3((G + per_block - 1) / per_block);
}
#define LAUNCH_CHECK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return az_fail("kernel launch", _e); } while (0)
#define DISPATCH_GAME(game, ...)                                   \
    do {                                                           \
        if ((game) == AZ_GAME_C4) { using GM = GameC4; __VA_ARGS__; } \
        else if ((game) == AZ_GAME_TTT) { using GM = GameTTT; __VA_ARGS__; } \
        else return az_fail("unknown game id");                       \
    } while (0)

// ---------------------------------------------------------------------------
// Scratch of az_c4_move_fused_static: the four per-game staging arrays that az_c4_move_fused takes from its
// caller (leaf positions, leaf words, priors, values passed between a workgroup's search steps and network
// tasks), for up to FUSED_STATIC_MAX_GAMES games, in device memory that belongs to the library image --
// nothing is allocated to use it.  Each GPU holds its own copy at its own address; host code only ever uses
// the per-device address looked up once in az_engine_create (the array's name is not a device pointer on the
// host side).
// ---------------------------------------------------------------------------
#define FUSED_STATIC_MAX_GAMES 16384
#define FS_OFF_BB   0                                               // uint64 [MAX][2]
#define FS_OFF_INFO (FS_OFF_BB + FUSED_STATIC_MAX_GAMES * 16)       // int32  [MAX]
#define FS_OFF_P    (FS_OFF_INFO + FUSED_STATIC_MAX_GAMES * 4)      // float  [MAX][7]
#define FS_OFF_V    (FS_OFF_P + FUSED_STATIC_MAX_GAMES * 28)        // float  [MAX]
#define FS_BYTES    (FS_OFF_V + FUSED_STATIC_MAX_GAMES * 4)         // 832 KiB
static __device__ __align__(256) uint8_t scratch[FS_BYTES];
#define MAX_DEVICES 64
static uint8_t* g_scratch_ptr[MAX_DEVICES];                         // device address of `scratch` per device, or null
static_assert(FS_OFF_INFO % 256 == 0 && FS_OFF_P % 256 == 0 && FS_OFF_V % 256 == 0, "scratch sub-arrays stay aligned");

static std::atomic<unsigned long long> g_fused_attr_done{0};        // bit d = k_c4_move_fused may use its LDS on device d
static std::atomic<unsigned long long> g_fused_hx_attr_done{0};     // the same for its fp16-pair form

// with device `dev` current: look the array's address up once per device, and give the kernel its dynamic-LDS
// attribute here, so that az_c4_move_fused_static itself is nothing but a launch
static int scratch_setup(int dev) {
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4_move_fused<false>), FUSED_LDS_BYTES, g_fused_attr_done)) return rc;
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4_move_fused<true>), FUSED_LDS_BYTES_HX, g_fused_hx_attr_done)) return rc;
    if (dev < 0 || dev >= MAX_DEVICES) return 0;                    // az_c4_move_fused_static refuses such a device
    if (g_scratch_ptr[dev]) return 0;
    void* ptr = nullptr;
    HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return az_fail("hipGetSymbolAddress(scratch) returned null");
    g_scratch_ptr[dev] = static_cast<uint8_t*>(ptr);
    return 0;
}

extern "C" {

const char* az_last_error(void) { return g_err.c_str(); }
int az_version(void) { return 1; }

static int engine_alloc(az_engine* e) {
    Dev& d = e->d;
    const size_t pool_bytes = (size_t)d.G * d.cap * (e->A + 1) * sizeof(uint4);
    hipError_t err;
    if ((err = hipMalloc(&d.pool, pool_bytes)) != hipSuccess ||
        (err = hipMalloc(&d.slots, (size_t)d.G * sizeof(Slot))) != hipSuccess ||
        (err = hipMalloc(&d.rec_visits, (size_t)d.G * AZ_MAX_MOVES * AZ_MAX_ACTIONS * 4)) != hipSuccess ||
        (err = hipMalloc(&d.ring, (size_t)d.ring_cap * sizeof(az_game_record))) != hipSuccess ||
        (err = hipMalloc(&d.ring_count, 512)) != hipSuccess ||
        (err = hipMalloc(&d.order, ((size_t)2 * d.G + 64) * 4)) != hipSuccess)
        return az_fail("hipMalloc (node pools)", err);
    HIPCHK(hipMemset(d.order, 0, ((size_t)2 * d.G + 64) * 4));
    HIPCHK(hipMemset(d.slots, 0, (size_t)d.G * sizeof(Slot)));
    HIPCHK(hipMemset(d.rec_visits, 0, (size_t)d.G * AZ_MAX_MOVES * AZ_MAX_ACTIONS * 4));
    HIPCHK(hipMemset(d.ring_count, 0, 512));
    int rc = az_reset_games(e, nullptr);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

int az_engine_create(const az_engine_config* cfg, az_engine** out) {
    if (!cfg || !out) return az_fail("null argument");
    *out = nullptr;
    if (cfg->game != AZ_GAME_C4 && cfg->game != AZ_GAME_TTT && cfg->game != AZ_GAME_GENERIC) return az_fail("unknown game id");
    if (cfg->num_games <= 0) return az_fail("num_games must be positive");
    if (cfg->num_simulations < 0) return az_fail("num_simulations must be >= 0");
    if (cfg->game == AZ_GAME_GENERIC && cfg->node_capacity <= 0)
        return az_fail("a user-defined game has no move bound: node_capacity must be given");
    const int lmax = cfg->game == AZ_GAME_C4 ? GameC4::LMAX : GameTTT::LMAX;
    long cap = cfg->node_capacity;
    if (cap <= 0) cap = cfg->reuse_search_tree ? (long)cfg->num_simulations * lmax + lmax + 1
                                               : (long)cfg->num_simulations + 2;
    if (cap >= (1l << 28)) return az_fail("node_capacity too large");
    DeviceScope scope(cfg->device);
    if (scope.err != hipSuccess) return az_fail("selecting the engine's device", scope.err);
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device));
    if (cus <= 0) return az_fail("device reports no compute units");
    if (int rc = scratch_setup(cfg->device)) return rc;
    az_engine* e = new az_engine();
    e->cfg = *cfg;
    e->A = cfg->game == AZ_GAME_C4 ? GameC4::A : cfg->game == AZ_GAME_TTT ? GameTTT::A : GEN_A;
    e->num_cus = cus;
    e->fused_wgs = 0;
    e->err_flags = 0;
    Dev& d = e->d;
    memset(&d, 0, sizeof(d));
    d.G = cfg->num_games; d.cap = (int)cap; d.S = cfg->num_simulations;
    d.explore_depth = cfg->exploration_depth; d.reuse = cfg->reuse_search_tree; d.version = cfg->model_version;
    d.alpha = cfg->alpha; d.eps = cfg->epsilon; d.c_base = cfg->c_base; d.c_init = cfg->c_init;
    d.one_minus_eps = (float)(1.0 - cfg->epsilon);
    d.seed = cfg->seed; d.first_uid = cfg->first_game_uid;
    d.uid_stride = cfg->uid_stride ? cfg->uid_stride : (uint64_t)cfg->num_games;
    d.ring_cap = 2u * (uint32_t)cfg->num_games + 16u;
    int rc = engine_alloc(e);
    if (!rc) {
        // A table for every visit sum a search of this engine can reach, from the host's libm; callers that need
        // the reference's bits (numpy's log) replace it with az_set_puct_lut.
        long reach = cfg->game == AZ_GAME_GENERIC ? cap + 2
                                                  : (long)cfg->num_simulations * (cfg->reuse_search_tree ? lmax : 1) + 2;
        if (reach < 1024) reach = 1024;
        if (reach > (1l << 24)) reach = 1l << 24;
        std::vector<double> t((size_t)2 * reach);
        for (long i = 0; i < reach; ++i) {
            t[2 * i] = log((1.0 + (double)i + cfg->c_base) / cfg->c_base) + cfg->c_init;
            t[2 * i + 1] = sqrt((double)i);
        }
        rc = az_set_puct_lut(e, t.data(), (int32_t)reach);
    }
    if (rc) {                         // nothing is handed out half-built: free what was allocated
        const std::string why = g_err;
        az_engine_destroy(e);
        g_err = why;
        return rc;
    }
    *out = e;
    return 0;
}

int az_engine_destroy(az_engine* e) {
    if (!e) return 0;
    DeviceScope scope(e->cfg.device);
    (void)hipFree(e->d.pool); (void)hipFree(e->d.slots); (void)hipFree(e->d.rec_visits); (void)hipFree(e->d.ring);
    (void)hipFree(e->d.ring_count); (void)hipFree(e->d.order); (void)hipFree(e->lut_dev); (void)hipFree(e->d.cache); (void)hipFree(e->d.cache_hits);
    delete e;
    return 0;
}

int az_reset_games(az_engine* e, void* stream) {
    if (!e) return az_fail("null engine");
    ENGINE_SCOPE(e);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(e->d.ring_count, 0, 512, st));
    e->err_flags = 0;
    // the reset touches slots only, whatever the game
    hipLaunchKernelGGL((k_reset<GameC4>), wave_grid(e->d.G), dim3(BLOCK_THREADS), 0, st, e->d);
    LAUNCH_CHECK();
    return 0;
}

int az_abandon_games(az_engine* e, void* stream) {
    if (!e) return az_fail("null engine");
    ENGINE_SCOPE(e);
    hipLaunchKernelGGL((k_abandon<GameC4>), wave_grid(e->d.G), dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d);
    LAUNCH_CHECK();
    return 0;
}

int az_set_puct_lut(az_engine* e, const double* lut_host, int32_t n) {
    if (!e || !lut_host || n <= 0) return az_fail("bad PUCT table");
    ENGINE_SCOPE(e);
    double2* dev = nullptr;
    HIPCHK(hipMalloc(&dev, (size_t)n * sizeof(double2)));
    HIPCHK(hipMemcpy(dev, lut_host, (size_t)n * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipDeviceSynchronize());
    (void)hipFree(e->lut_dev);
    e->lut_dev = dev; e->d.lut = dev; e->d.lut_len = n;
    return 0;
}

int az_set_model_version(az_engine* e, int32_t version) {
    if (!e) return az_fail("null engine");
    e->d.version = version; e->cfg.model_version = version;
    return 0;
}

// hit counters: one per wavefront of any search grid (G / 8 trees per wave at least, at most 16 waves per
// workgroup of the one-launch-per-move kernel, whose grid never exceeds G workgroups)
static size_t cache_counters(int G) { return (size_t)16 * G + 16; }

int az_eval_cache_enable(az_engine* e, int32_t log2_entries) {
    if (!e) return az_fail("null engine");
    ENGINE_SCOPE(e);
    if (log2_entries != 0 && (log2_entries < 10 || log2_entries > 28)) return az_fail("eval cache: log2_entries must be 0 or 10..28");
    HIPCHK(hipDeviceSynchronize());
    (void)hipFree(e->d.cache); (void)hipFree(e->d.cache_hits);
    e->d.cache = nullptr; e->d.cache_mask = 0; e->d.cache_hits = nullptr;
    if (log2_entries == 0) return 0;
    const size_t bytes = ((size_t)1 << log2_entries) * 64;
    const size_t counters = cache_counters(e->d.G);
    uint4* t = nullptr;
    unsigned long long* c = nullptr;
    hipError_t err = hipMalloc(&t, bytes);
    if (err == hipSuccess) err = hipMalloc(&c, counters * 8);
    if (err != hipSuccess) { (void)hipFree(t); return az_fail("hipMalloc (evaluation cache)", err); }
    HIPCHK(hipMemset(t, 0, bytes));
    HIPCHK(hipMemset(c, 0, counters * 8));
    HIPCHK(hipDeviceSynchronize());
    e->d.cache = t; e->d.cache_mask = (uint32_t)(((size_t)1 << log2_entries) - 1); e->d.cache_hits = c;
    return 0;
}

int az_eval_cache_hits(az_engine* e, uint64_t* hits, void* stream) {
    if (!e || !hits) return az_fail("null argument");
    ENGINE_SCOPE(e);
    *hits = 0;
    if (!e->d.cache_hits) return 0;
    std::vector<unsigned long long> h(cache_counters(e->d.G));
    HIPCHK(hipMemcpyAsync(h.data(), e->d.cache_hits, h.size() * 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (unsigned long long x : h) *hits += x;
    return 0;
}

int az_begin_move(az_engine* e, uint64_t* leaf_bb, int32_t* leaf_info, void* stream) {
    if (!e || !leaf_bb || !leaf_info) return az_fail("null argument");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_begin_move<GM, false>), wave_grid(e->d.G), dim3(BLOCK_THREADS), 0,
                                                  (hipStream_t)stream, e->d, leaf_bb, leaf_info, (const double*)nullptr));
    LAUNCH_CHECK();
    return 0;
}

int az_begin_move_uniform(az_engine* e, const double* eta, void* stream) {
    if (!e) return az_fail("null engine");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_begin_move<GM, true>), wave_grid(e->d.G), dim3(BLOCK_THREADS), 0,
                                                  (hipStream_t)stream, e->d, (uint64_t*)nullptr, (int32_t*)nullptr, eta));
    LAUNCH_CHECK();
    return 0;
}

int az_apply_root(az_engine* e, const float* p, const double* eta, void* stream) {
    if (!e || !p) return az_fail("null argument");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_apply_root<GM>), wave_grid(e->d.G), dim3(BLOCK_THREADS), 0,
                                                  (hipStream_t)stream, e->d, p, eta));
    LAUNCH_CHECK();
    return 0;
}

int az_select(az_engine* e, uint64_t* leaf_bb, int32_t* leaf_info, void* stream) {
    if (!e || !leaf_bb || !leaf_info) return az_fail("null argument");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, { const int gs_ = group_size_for(GroupSize<GM>::min_value, e->d.G); const dim3 gr_ = group_grid(e->d.G, gs_);
  switch (gs_) {
    case 8: if constexpr (8 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 8, false, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (const float*)nullptr, (const float*)nullptr, leaf_bb, leaf_info); break;
    case 16: if constexpr (16 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 16, false, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (const float*)nullptr, (const float*)nullptr, leaf_bb, leaf_info); break;
    case 32: if constexpr (32 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 32, false, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (const float*)nullptr, (const float*)nullptr, leaf_bb, leaf_info); break;
    case 64: if constexpr (64 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 64, false, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (const float*)nullptr, (const float*)nullptr, leaf_bb, leaf_info); break;
  } });
    LAUNCH_CHECK();
    return 0;
}

int az_expand_backup(az_engine* e, const float* p, const float* v, void* stream) {
    if (!e || !p || !v) return az_fail("null argument");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, { const int gs_ = group_size_for(GroupSize<GM>::min_value, e->d.G); const dim3 gr_ = group_grid(e->d.G, gs_);
  switch (gs_) {
    case 8: if constexpr (8 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 8, true, false>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, (uint64_t*)nullptr, (int32_t*)nullptr); break;
    case 16: if constexpr (16 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 16, true, false>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, (uint64_t*)nullptr, (int32_t*)nullptr); break;
    case 32: if constexpr (32 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 32, true, false>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, (uint64_t*)nullptr, (int32_t*)nullptr); break;
    case 64: if constexpr (64 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 64, true, false>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, (uint64_t*)nullptr, (int32_t*)nullptr); break;
  } });
    LAUNCH_CHECK();
    return 0;
}

int az_sim_step(az_engine* e, float* p, float* v, uint64_t* leaf_bb, int32_t* leaf_info, void* stream) {
    if (!e || !leaf_bb || !leaf_info) return az_fail("null argument");
    ENGINE_SCOPE(e);
    if (!p) return az_select(e, leaf_bb, leaf_info, stream);
    if (!v) return az_fail("v is required with p");
    DISPATCH_GAME(e->cfg.game, { const int gs_ = group_size_for(GroupSize<GM>::min_value, e->d.G); const dim3 gr_ = group_grid(e->d.G, gs_);
  switch (gs_) {
    case 8: if constexpr (8 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 8, true, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, leaf_bb, leaf_info); break;
    case 16: if constexpr (16 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 16, true, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, leaf_bb, leaf_info); break;
    case 32: if constexpr (32 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 32, true, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, leaf_bb, leaf_info); break;
    case 64: if constexpr (64 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_sim_step_p<GM, 64, true, true>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, p, v, leaf_bb, leaf_info); break;
  } });
    LAUNCH_CHECK();
    return 0;
}

#ifdef AZ_EXP_TIMING
int az_debug_workgroups(unsigned long long* out2048) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_wg_dbg), sizeof(unsigned long long) * 2048));
    return 0;
}
int az_debug_waits(unsigned long long* out10) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out10, HIP_SYMBOL(g_wait_dbg), 80));
    unsigned long long z[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wait_dbg), z, 80));
    return 0;
}
int az_debug_done(unsigned long long* out8) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_done_dbg), 64));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_done_dbg), z, 64));
    return 0;
}
// out: the waves' cycles in network tasks / search steps / idle and the number of waves (k_c4_move_fused), then
// eight more counters of that kernel: two-board tasks, one-board tasks, search steps, games stepped, the slowest
// wave's cycles, cycles in head batches, cycles in hand-overs, head batches
int az_debug_counters(az_engine* e, unsigned long long* out12) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out12, e->d.ring_count + 16, 32, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->d.ring_count + 16, 0, 32));
    HIPCHK(hipMemcpyFromSymbol(out12 + 4, HIP_SYMBOL(g_net_dbg), 64));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_net_dbg), z, 64));
    return 0;
}
#endif

#ifdef AZ_EXP_ASSERT
// diagnostic builds: the first failed index check {code, value, workgroup, thread}, zeros if none
int az_debug_assert_words(az_engine* e, uint32_t* out4) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out4, e->d.ring_count + 24, 16, hipMemcpyDeviceToHost));
    return 0;
}
#endif

// games per workgroup of the one-launch-per-move kernel: spread over one workgroup per compute unit, at most
// 64 (eight search lanes per game)
#define FUSED_MAX_GPW 64
static int fused_workgroups(const az_engine* e) {
    return (e->fused_wgs > 0 && e->fused_wgs < e->num_cus) ? e->fused_wgs : e->num_cus;
}
static int fused_launch(az_engine* e, const uint8_t* blob, int32_t num_blocks, uint64_t* leaf_bb, int32_t* leaf_info,
                        float* p, float* v, hipStream_t st, bool hx = false) {
    if (e->cfg.game != AZ_GAME_C4) return az_fail("az_c4_move_fused: Connect Four only");
    if (num_blocks < 0) return az_fail("az_c4_move_fused: num_blocks");
    if (e->d.S <= 0) return 0;
    const int wgs = fused_workgroups(e);
    const int gpw = (e->d.G + wgs - 1) / wgs;
    if (gpw > FUSED_MAX_GPW) return az_fail("az_c4_move_fused: more than 64 games per workgroup");
    const int grid = (e->d.G + gpw - 1) / gpw;
    // already done by az_engine_create on this device (a bitmap test, then nothing)
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4_move_fused<false>), FUSED_LDS_BYTES, g_fused_attr_done)) return rc;
    if (int rc = az_set_kernel_lds(reinterpret_cast<const void*>(k_c4_move_fused<true>), FUSED_LDS_BYTES_HX, g_fused_hx_attr_done)) return rc;
    if (e->d.G > BALANCE_MAX_GAMES) return az_fail("az_c4_move_fused: more than 16384 games");
    hipLaunchKernelGGL(k_fused_balance, dim3(1), dim3(1024), 0, st, e->d, gpw, grid);
    if (hx)
        hipLaunchKernelGGL(k_c4_move_fused<true>, dim3(grid), dim3(FUSED_THREADS), FUSED_LDS_BYTES_HX, st, e->d, blob, (int)num_blocks,
                           leaf_bb, leaf_info, p, v, (int)e->d.S, gpw);
    else
        hipLaunchKernelGGL(k_c4_move_fused<false>, dim3(grid), dim3(FUSED_THREADS), FUSED_LDS_BYTES, st, e->d, blob, (int)num_blocks,
                           leaf_bb, leaf_info, p, v, (int)e->d.S, gpw);
    AZ_LAUNCH_CHECK("k_c4_move_fused");
    return 0;
}

int az_c4_move_fused_max_games(const az_engine* e) { return e ? FUSED_MAX_GPW * fused_workgroups(e) : -1; }

int az_c4_move_fused_limit_workgroups(az_engine* e, int32_t max_workgroups) {
    if (!e || max_workgroups < 0) return az_fail("az_c4_move_fused_limit_workgroups: bad argument");
    e->fused_wgs = max_workgroups;
    return 0;
}

int az_c4_move_fused_counters(az_engine* e, uint64_t* out4, void* stream) {
    if (!e || !out4) return az_fail("null argument");
    ENGINE_SCOPE(e);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(out4, e->d.ring_count + FUSED_COUNTER_WORD, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(e->d.ring_count + FUSED_COUNTER_WORD, 0, 32, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}

int az_c4_move_fused(az_engine* e, const uint8_t* blob, int32_t num_blocks, uint64_t* leaf_bb, int32_t* leaf_info,
                     float* p, float* v, void* stream) {
    if (!e || !blob || !leaf_bb || !leaf_info || !p || !v) return az_fail("null argument");
    ENGINE_SCOPE(e);
    return fused_launch(e, blob, num_blocks, leaf_bb, leaf_info, p, v, (hipStream_t)stream);
}

int az_c4_move_fused_hx(az_engine* e, const uint8_t* blob, int32_t num_blocks, uint64_t* leaf_bb, int32_t* leaf_info,
                        float* p, float* v, void* stream) {
    if (!e || !blob || !leaf_bb || !leaf_info || !p || !v) return az_fail("null argument");
    ENGINE_SCOPE(e);
    return fused_launch(e, blob, num_blocks, leaf_bb, leaf_info, p, v, (hipStream_t)stream, true);
}

int az_c4_move_fused_static(az_engine* e, const uint8_t* blob, int32_t num_blocks, void* stream) {
    if (!e || !blob) return az_fail("null argument");
    ENGINE_SCOPE(e);
    if (e->d.G > FUSED_STATIC_MAX_GAMES)
        return az_fail("az_c4_move_fused_static: more than 16384 games (the size of the built-in staging area)");
    const int dev = e->cfg.device;
    uint8_t* base = (dev >= 0 && dev < MAX_DEVICES) ? g_scratch_ptr[dev] : nullptr;
    if (!base) return az_fail("az_c4_move_fused_static: no staging area on this device (ordinal beyond 63)");
    return fused_launch(e, blob, num_blocks, reinterpret_cast<uint64_t*>(base + FS_OFF_BB),
                        reinterpret_cast<int32_t*>(base + FS_OFF_INFO), reinterpret_cast<float*>(base + FS_OFF_P),
                        reinterpret_cast<float*>(base + FS_OFF_V), (hipStream_t)stream);
}

int az_run_sims_uniform(az_engine* e, int32_t num_sims, void* stream) {
    if (!e || num_sims < 0) return az_fail("bad argument");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, { const int gs_ = group_size_for(GroupSize<GM>::min_value, e->d.G); const dim3 gr_ = group_grid(e->d.G, gs_);
  switch (gs_) {
    case 8: if constexpr (8 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_run_uniform_p<GM, 8>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (int)num_sims); break;
    case 16: if constexpr (16 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_run_uniform_p<GM, 16>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (int)num_sims); break;
    case 32: if constexpr (32 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_run_uniform_p<GM, 32>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (int)num_sims); break;
    case 64: if constexpr (64 >= GroupSize<GM>::min_value) hipLaunchKernelGGL((k_run_uniform_p<GM, 64>), gr_, dim3(BLOCK_THREADS), 0, (hipStream_t)stream, e->d, (int)num_sims); break;
  } });
    LAUNCH_CHECK();
    return 0;
}

int az_choose_and_advance(az_engine* e, const double* u, int32_t* action, int32_t* num_edges, uint32_t* visits,
                          int32_t* winner, void* stream) {
    if (!e) return az_fail("null engine");
    ENGINE_SCOPE(e);
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_choose<GM>), wave_grid(e->d.G), dim3(BLOCK_THREADS), 0,
                                                  (hipStream_t)stream, e->d, u, action, num_edges, visits, winner));
    LAUNCH_CHECK();
    return 0;
}

int az_drain_finished(az_engine* e, az_game_record* out_host, int32_t max_records, int32_t* num_out, void* stream) {
    if (!e || !num_out) return az_fail("null argument");
    ENGINE_SCOPE(e);
    hipStream_t st = (hipStream_t)stream;
    uint32_t head[2] = {0, 0};               // {records in the ring, error bits}
    HIPCHK(hipMemcpyAsync(head, e->d.ring_count, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint32_t count = head[0];
    e->err_flags |= head[1];
    if (count > e->d.ring_cap) count = e->d.ring_cap;
    if ((int32_t)count > max_records) return az_fail("az_drain_finished: output buffer too small");
    if (count && !out_host) return az_fail("null output buffer");
    if (count) HIPCHK(hipMemcpyAsync(out_host, e->d.ring, (size_t)count * sizeof(az_game_record), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemsetAsync(e->d.ring_count, 0, 4, st));
    HIPCHK(hipStreamSynchronize(st));
    *num_out = (int32_t)count;
    return 0;
}

int az_take_finished_device(az_engine* e, az_game_record* out_dev, int32_t max_records, int32_t* head_dev, void* stream) {
    if (!e || !out_dev || !head_dev || max_records < 0) return az_fail("bad argument");
    ENGINE_SCOPE(e);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_take_finished, dim3(256), dim3(256), 0, st, e->d.ring, e->d.ring_count, e->d.ring_cap, out_dev,
                       (uint32_t)max_records, head_dev);
    AZ_LAUNCH_CHECK("k_take_finished");
    HIPCHK(hipMemsetAsync(e->d.ring_count, 0, 4, st));
    return 0;
}

int az_engine_error_flags(const az_engine* e) { return e ? (int)e->err_flags : -1; }

int az_get_stats(az_engine* e, az_stats* out, void* stream) {
    if (!e || !out) return az_fail("null argument");
    ENGINE_SCOPE(e);
    hipStream_t st = (hipStream_t)stream;
    std::vector<Slot> h((size_t)e->d.G);
    HIPCHK(hipMemcpyAsync(h.data(), e->d.slots, h.size() * sizeof(Slot), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memset(out, 0, sizeof(*out));
    for (const Slot& s : h) {
        out->sims += s.st_sims; out->select_steps += s.st_steps; out->leaf_evals += s.st_evals;
        out->root_evals += s.st_root_evals; out->moves += s.st_moves; out->games += s.st_games;
        out->errors += s.err ? 1 : 0;
    }
    return 0;
}

// ---- user-defined games: tree on the device, rules on the host ----------------------------------------------
static int generic_check(az_engine* e, int32_t g) {
    if (e->cfg.game != AZ_GAME_GENERIC) return az_fail("this call is for engines created with AZ_GAME_GENERIC");
    if (g < 0 || g >= e->d.G) return az_fail("bad game slot");
    return 0;
}

int az_generic_root(az_engine* e, int32_t g, int32_t n_edges, const float* p_host, const double* eta_host, void* stream) {
    ENGINE_SCOPE(e);
    if (int rc = generic_check(e, g)) return rc;
    if (n_edges < 0 || n_edges > GEN_A) return az_fail("az_generic_root: at most 15 actions");
    if (n_edges > 0 && !p_host) return az_fail("az_generic_root: priors missing");
    GenPrior pr; GenNoise no;
    memset(&pr, 0, sizeof(pr)); memset(&no, 0, sizeof(no));
    for (int i = 0; i < n_edges; ++i) pr.p[i] = p_host[i];
    if (eta_host) for (int i = 0; i < GEN_A; ++i) no.eta[i] = eta_host[i];
    hipLaunchKernelGGL(k_generic_root, dim3(1), dim3(64), 0, (hipStream_t)stream, e->d, (int)g, (int)n_edges, pr, no,
                       eta_host ? 1 : 0);
    AZ_LAUNCH_CHECK("k_generic_root");
    return 0;
}

int az_generic_select(az_engine* e, int32_t g, int32_t* depth_host, uint8_t* slots_host, void* stream) {
    ENGINE_SCOPE(e);
    if (int rc = generic_check(e, g)) return rc;
    if (!depth_host || !slots_host) return az_fail("null argument");
    hipStream_t st = (hipStream_t)stream;
    uint32_t* out = reinterpret_cast<uint32_t*>(e->d.ring_count) + 64;  // the second half of the 512-byte counter block
    hipLaunchKernelGGL(k_generic_select, dim3(1), dim3(64), 0, st, e->d, (int)g, out);
    AZ_LAUNCH_CHECK("k_generic_select");
    uint32_t host[1 + PATH_LEN];
    HIPCHK(hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (host[0] > PATH_LEN) {
        e->err_flags |= 32u;
        *depth_host = 0;
        return az_fail("az_generic_select: the search path is deeper than 48 levels (the tree is unchanged; search from a node nearer the leaves)");
    }
    *depth_host = (int32_t)host[0];
    for (uint32_t d = 0; d < host[0]; ++d) slots_host[d] = (uint8_t)host[1 + d];
    return 0;
}

int az_generic_expand_backup(az_engine* e, int32_t g, int32_t n_edges, const float* p_host, float v_edge, void* stream) {
    ENGINE_SCOPE(e);
    if (int rc = generic_check(e, g)) return rc;
    if (n_edges < 0 || n_edges > GEN_A) return az_fail("az_generic_expand_backup: at most 15 actions");
    if (n_edges > 0 && !p_host) return az_fail("az_generic_expand_backup: priors missing");
    GenPrior pr;
    memset(&pr, 0, sizeof(pr));
    for (int i = 0; i < n_edges; ++i) pr.p[i] = p_host[i];
    hipLaunchKernelGGL(k_generic_expand_backup, dim3(1), dim3(64), 0, (hipStream_t)stream, e->d, (int)g, (int)n_edges, pr,
                       v_edge);
    AZ_LAUNCH_CHECK("k_generic_expand_backup");
    return 0;
}

int az_node_words(const az_engine* e) { return e ? 4 * (e->A + 1) : 0; }

int az_set_position(az_engine* e, int32_t g, uint64_t bb_next, uint64_t bb_prev, int32_t num_moves,
                    const uint32_t* nodes_host, int32_t num_nodes, uint32_t root_node) {
    if (!e || g < 0 || g >= e->d.G) return az_fail("bad game slot");
    ENGINE_SCOPE(e);
    if (num_nodes < 0 || num_nodes > e->d.cap) return az_fail("tree larger than node_capacity");
    if (root_node != AZ_NO_NODE && root_node >= (uint32_t)num_nodes) return az_fail("root_node out of range");
    HIPCHK(hipDeviceSynchronize());
    Slot s;
    HIPCHK(hipMemcpy(&s, e->d.slots + g, sizeof(Slot), hipMemcpyDeviceToHost));
    s.nx = bb_next; s.pv = bb_prev; s.nm = num_moves; s.root = root_node; s.alloc = (uint32_t)num_nodes;
    s.leaf_kind = AZ_LEAF_NONE; s.leaf_depth = 0; s.move_sims = 0; s.err = 0;
    HIPCHK(hipMemcpy(e->d.slots + g, &s, sizeof(Slot), hipMemcpyHostToDevice));
    const size_t node_bytes = (size_t)(e->A + 1) * sizeof(uint4);
    if (num_nodes) {
        if (!nodes_host) return az_fail("null node image");
        // child indices must stay inside the image
        const int words = 4 * (e->A + 1);
        for (int n = 0; n < num_nodes; ++n) {
            const uint32_t ne = nodes_host[(size_t)n * words + 4 * e->A];
            if (ne > (uint32_t)e->A) return az_fail("node image: num_edges out of range");
            for (uint32_t i = 0; i < ne; ++i) {
                const uint32_t c = nodes_host[(size_t)n * words + 4 * i + 3];
                if (c != AZ_NO_NODE && c >= (uint32_t)num_nodes) return az_fail("node image: child out of range");
            }
        }
        HIPCHK(hipMemcpy((char*)e->d.pool + (size_t)g * e->d.cap * node_bytes, nodes_host, (size_t)num_nodes * node_bytes,
                         hipMemcpyHostToDevice));
    }
    return 0;
}

int az_get_tree(az_engine* e, int32_t g, uint32_t* nodes_host, int32_t max_nodes, int32_t* num_nodes,
                uint32_t* root_node) {
    if (!e || g < 0 || g >= e->d.G || !num_nodes || !root_node) return az_fail("bad argument");
    ENGINE_SCOPE(e);
    HIPCHK(hipDeviceSynchronize());
    Slot s;
    HIPCHK(hipMemcpy(&s, e->d.slots + g, sizeof(Slot), hipMemcpyDeviceToHost));
    uint32_t flags = 0;               // this call has synchronised anyway: bring az_engine_error_flags up to date
    HIPCHK(hipMemcpy(&flags, e->d.ring_count + 1, 4, hipMemcpyDeviceToHost));
    e->err_flags |= flags | s.err;
    *num_nodes = (int32_t)s.alloc; *root_node = s.root;
    if (!nodes_host) return 0;
    if ((int32_t)s.alloc > max_nodes) return az_fail("az_get_tree: buffer too small");
    const size_t node_bytes = (size_t)(e->A + 1) * sizeof(uint4);
    if (s.alloc)
        HIPCHK(hipMemcpy(nodes_host, (char*)e->d.pool + (size_t)g * e->d.cap * node_bytes, (size_t)s.alloc * node_bytes,
                         hipMemcpyDeviceToHost));
    return 0;
}

// ---- stateless bulk kernels -------------------------------------------------
int az_positions_from_history(int32_t game, const uint8_t* hist, const int32_t* len, int32_t stride, int32_t n,
                              uint64_t* bb, int32_t* num_moves, void* stream) {
    if (n < 0 || (n && (!hist || !len || !bb || !num_moves))) return az_fail("bad argument");
    if (n == 0) return 0;
    DISPATCH_GAME(game, hipLaunchKernelGGL((k_from_history<GM>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                                           hist, len, (int)stride, (int)n, bb, num_moves));
    LAUNCH_CHECK();
    return 0;
}

int az_game_query(int32_t game, const uint64_t* bb, const int32_t* num_moves, int32_t n, int32_t* legal_mask,
                  int32_t* terminal, int32_t* winner, void* stream) {
    if (n < 0 || (n && (!bb || !num_moves))) return az_fail("bad argument");
    if (n == 0) return 0;
    DISPATCH_GAME(game, hipLaunchKernelGGL((k_query<GM>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, bb,
                                           num_moves, (int)n, legal_mask, terminal, winner));
    LAUNCH_CHECK();
    return 0;
}

int az_game_step(int32_t game, uint64_t* bb, int32_t* num_moves, const int32_t* action, int32_t n, int32_t* status,
                 void* stream) {
    if (n < 0 || (n && (!bb || !num_moves || !action))) return az_fail("bad argument");
    if (n == 0) return 0;
    DISPATCH_GAME(game, hipLaunchKernelGGL((k_step<GM>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, bb,
                                           num_moves, action, (int)n, status));
    LAUNCH_CHECK();
    return 0;
}

int az_render(int32_t game, const uint64_t* bb, int32_t n, float* x, uint8_t* valid, void* stream) {
    if (n < 0 || (n && !bb)) return az_fail("bad argument");
    if (n == 0) return 0;
    DISPATCH_GAME(game, {
        const long total = (long)n * 2 * GM::H * GM::W;
        hipLaunchKernelGGL((k_render<GM>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bb,
                           (int)n, x, valid);
    });
    LAUNCH_CHECK();
    return 0;
}

int az_examples_from_records(int32_t game, const az_game_record* records, int32_t n, const int32_t* state_offset,
                             int32_t total, float* x, float* p, float* z, uint8_t* valid, int32_t* version,
                             void* stream) {
    if (n < 0 || total < 0 || (total && (!records || !state_offset || !x || !p || !z || !valid))) return az_fail("bad argument");
    if (total == 0) return 0;
    DISPATCH_GAME(game, hipLaunchKernelGGL((k_examples<GM>), dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                                           records, (int)n, state_offset, (int)total, x, p, z, valid, version));
    LAUNCH_CHECK();
    return 0;
}

int az_masked_softmax(int32_t game, const float* logits, const int32_t* leaf_info, int32_t n, float* p, void* stream) {
    if (n < 0 || (n && (!logits || !leaf_info || !p))) return az_fail("bad argument");
    if (n == 0) return 0;
    DISPATCH_GAME(game, hipLaunchKernelGGL((k_masked_softmax<GM::A>), dim3((n + 255) / 256), dim3(256), 0,
                                           (hipStream_t)stream, logits, leaf_info, (int)n, p));
    LAUNCH_CHECK();
    return 0;
}

int az_ucb_score(az_engine* e, const uint32_t* N, const float* W, const float* P, const int32_t* num_edges, int32_t n,
                 double* score, int32_t* argmax, void* stream) {
    if (!e || n < 0 || (n && (!N || !W || !P || !num_edges || !score || !argmax))) return az_fail("bad argument");
    ENGINE_SCOPE(e);
    if (n == 0) return 0;
    DISPATCH_GAME(e->cfg.game, hipLaunchKernelGGL((k_ucb<GM::A>), dim3((n + 255) / 256), dim3(256), 0,
                                                  (hipStream_t)stream, e->d, N, W, P, num_edges, (int)n, score, argmax));
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
