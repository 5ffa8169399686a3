// knn.hip -- exact K nearest neighbours of a point cloud (r3dgs_knn, include/r3dgs_reduction.h); replaces
// SimpleKNN::knn / knn_index2 of /root/reference/submodules/simple-knn/simple_knn.cu:179-224, :468-513.
//
// What the reference computes: per point, the K smallest squared distances to the OTHER points (by index) and
// their indices; `knn` returns the mean of the 3 smallest.  How it gets there (one thread per point walking
// 128-point Morton boxes outward, K-best lists read-modify-written in global memory, three host syncs for the
// bounding box) is not reproduced.  Here:
//   1. bounding box by a block reduce + ordered-int atomics, no host read-back;
//   2. 30-bit Morton codes, rocPRIM radix sort of (code, index);
//   3. points gathered into sorted order as float4 {x,y,z,index}; one AABB per 64 consecutive points (= one wave);
//   4. one WAVE per 64 consecutive queries.  Each lane keeps its K-best list as a max-heap in an LDS column (bank =
//      lane, so the accesses are conflict-free whatever path a lane's sift takes).  The wave first scans its own and adjacent boxes (seed), then every lane
//      tests a different box's AABB against the wave's AABB and current worst distance (64 box tests per step);
//      surviving boxes are staged through LDS and scored by all lanes with broadcast reads in a divergence-free
//      pass that only records which points beat the lane's worst entry; the (rare) insertions follow per lane.
// The result is canonical: ascending by (distance, index), ties at the K-th place resolved by the smaller index.
// Compiled without FMA contraction so that the box lower bounds (monotone in every rounding step) can never exceed
// a contained point's distance computed by the same expression: the search is exact in fp32.
#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/r3dgs_reduction.h"
#include "common.h"

namespace {

constexpr int kWave = 64;
constexpr int kSparse = 12;     // <= this many wanting queries: score a box with lanes = points instead of lanes = queries
constexpr int kMaxK = 112;      // LDS: K * 64 lanes * 8 B + 1 KB staging <= 64 KB per wave
constexpr float kFltMax = 3.402823466e+38f;

struct KnnWork {
    int* bbox;            // 6 ordered-int encoded floats: min xyz, max xyz
    uint32_t* code;       // [P]
    uint32_t* code_sorted;
    uint32_t* index;      // [P] iota
    uint32_t* index_sorted;
    float4* sorted;       // [P] {x,y,z,as_float(index)} in Morton order
    float4* box_min;      // [nb]
    float4* box_max;      // [nb]
    char* temp;
    static KnnWork carve(char* base, size_t P, size_t temp_bytes)
    {
        KnnWork w;
        char* p = base;
        auto take = [&](size_t bytes) {
            char* r = p;
            p += (bytes + 255) / 256 * 256;
            return r;
        };
        const size_t nb = (P + kWave - 1) / kWave;
        w.bbox = reinterpret_cast<int*>(take(256));
        w.code = reinterpret_cast<uint32_t*>(take(4 * P));
        w.code_sorted = reinterpret_cast<uint32_t*>(take(4 * P));
        w.index = reinterpret_cast<uint32_t*>(take(4 * P));
        w.index_sorted = reinterpret_cast<uint32_t*>(take(4 * P));
        w.sorted = reinterpret_cast<float4*>(take(16 * P));
        w.box_min = reinterpret_cast<float4*>(take(16 * nb));
        w.box_max = reinterpret_cast<float4*>(take(16 * nb));
        w.temp = take(temp_bytes);
        (void)p;
        return w;
    }
};

size_t sort_temp_bytes(size_t P)
{
    size_t bytes = 0;
    R3_HIP(rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, P, 0, 30, (hipStream_t)0));
    return bytes;
}

// float <-> int whose signed order matches the float order (so atomicMin/atomicMax work on floats)
__device__ inline int f2ord(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ inline float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ inline float wave_min(float v)
{
    for (int off = 32; off; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ inline float wave_max(float v)
{
    for (int off = 32; off; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__global__ void bbox_init_kernel(int* bbox)
{
    if (threadIdx.x < 3) bbox[threadIdx.x] = f2ord(kFltMax);
    else if (threadIdx.x < 6) bbox[threadIdx.x] = f2ord(-kFltMax);
}

__global__ __launch_bounds__(256) void bbox_kernel(int P, const float* __restrict__ pts, int* __restrict__ bbox)
{
    __shared__ float s_red[4][6];
    float mn[3] = {kFltMax, kFltMax, kFltMax}, mx[3] = {-kFltMax, -kFltMax, -kFltMax};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    for (int a = 0; a < 3; a++) {
        mn[a] = wave_min(mn[a]);
        mx[a] = wave_max(mx[a]);
    }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) {
            s_red[threadIdx.x >> 6][a] = mn[a];
            s_red[threadIdx.x >> 6][3 + a] = mx[a];
        }
    __syncthreads();
    if (threadIdx.x < 6) {              // one atomic per block and component (same-address atomics serialise)
        const int a = threadIdx.x;
        float v = s_red[0][a];
        for (int w = 1; w < 4; w++) v = a < 3 ? fminf(v, s_red[w][a]) : fmaxf(v, s_red[w][a]);
        if (a < 3) atomicMin(&bbox[a], f2ord(v));
        else atomicMax(&bbox[a], f2ord(v));
    }
}

__device__ inline uint32_t spread10(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void morton_kernel(int P, const float* __restrict__ pts, const int* __restrict__ bbox,
                                                     uint32_t* __restrict__ code, uint32_t* __restrict__ index)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(bbox[a]), hi = ord2f(bbox[3 + a]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (pts[3 * i + a] - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);            // also maps NaN to 0
        q[a] = (uint32_t)(t * 1023.f);
    }
    code[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    index[i] = (uint32_t)i;
}

// one wave per box: gather its 64 points into sorted order and reduce their AABB
__global__ __launch_bounds__(256) void gather_boxes_kernel(int P, const float* __restrict__ pts,
                                                           const uint32_t* __restrict__ index_sorted,
                                                           float4* __restrict__ sorted, float4* __restrict__ box_min,
                                                           float4* __restrict__ box_max)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < P;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
        const uint32_t id = index_sorted[i];
        x = pts[3 * id];
        y = pts[3 * id + 1];
        z = pts[3 * id + 2];
        sorted[i] = make_float4(x, y, z, __uint_as_float(id));
    }
    const float mnx = wave_min(live ? x : kFltMax), mny = wave_min(live ? y : kFltMax), mnz = wave_min(live ? z : kFltMax);
    const float mxx = wave_max(live ? x : -kFltMax), mxy = wave_max(live ? y : -kFltMax),
                mxz = wave_max(live ? z : -kFltMax);
    if ((threadIdx.x & 63) == 0 && live) {
        box_min[i >> 6] = make_float4(mnx, mny, mnz, 0.f);
        box_max[i >> 6] = make_float4(mxx, mxy, mxz, 0.f);
    }
}

__device__ inline float gap(float lo_a, float hi_a, float lo_b, float hi_b)
{
    // distance between the intervals [lo_a,hi_a] and [lo_b,hi_b] (0 if they overlap)
    return fmaxf(0.f, fmaxf(lo_b - hi_a, lo_a - hi_b));
}

// K-best list of one lane: a max-heap by (distance, index) in an LDS column (slot j at d[j * 64]); the root is the
// current worst entry and is mirrored in registers.
struct Best {
    float* d;
    int* id;
    int K;
    float wd;
    int wid;

    __device__ inline void reset()
    {
        for (int j = 0; j < K; j++) {
            d[j * kWave] = kFltMax;
            id[j * kWave] = 0x7fffffff;   // sorts after every real index; rewritten to -1 on output
        }
        wd = kFltMax;
        wid = 0x7fffffff;
    }
    __device__ inline bool better(float cd, int cid) const { return cd < wd || (cd == wd && cid < wid); }
    static __device__ inline bool greater(float ad, int ai, float bd, int bi) { return ad > bd || (ad == bd && ai > bi); }
    // replace the root by (cd, cid) -- the caller checked better() -- and sift it down
    __device__ inline void insert(float cd, int cid)
    {
        int i = 0;
        for (;;) {
            const int l = 2 * i + 1;
            if (l >= K) break;
            int c = l;
            float dc = d[l * kWave];
            int ic = id[l * kWave];
            if (l + 1 < K) {
                const float dr = d[(l + 1) * kWave];
                const int ir = id[(l + 1) * kWave];
                if (greater(dr, ir, dc, ic)) {
                    c = l + 1;
                    dc = dr;
                    ic = ir;
                }
            }
            if (!greater(dc, ic, cd, cid)) break;
            d[i * kWave] = dc;
            id[i * kWave] = ic;
            i = c;
        }
        d[i * kWave] = cd;
        id[i * kWave] = cid;
        wd = d[0];
        wid = id[0];
    }
};

// MEAN3: write (d0 + d1 + d2) / 3 instead of the lists (K == 3).
template <bool MEAN3>
__global__ __launch_bounds__(kWave) void knn_kernel(int P, int K, const float4* __restrict__ sorted,
                                                    const float4* __restrict__ box_min,
                                                    const float4* __restrict__ box_max, float* __restrict__ dists,
                                                    int* __restrict__ indices, float* __restrict__ mean3)
{
    extern __shared__ char smem[];
    float4* s_cand = reinterpret_cast<float4*>(smem);                        // 64 staged candidates
    float* s_d = reinterpret_cast<float*>(smem + kWave * sizeof(float4));    // [K][64]
    int* s_id = reinterpret_cast<int*>(s_d + (size_t)K * kWave);             // [K][64]

    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int nb = (P + kWave - 1) / kWave;
    const int qi = b * kWave + lane;
    const bool live = qi < P;
    const float4 q = live ? sorted[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int my_id = live ? (int)__float_as_uint(q.w) : -1;
    const float4 wmin = box_min[b], wmax = box_max[b];

    Best best{s_d + lane, s_id + lane, K, 0.f, 0};
    best.reset();

    auto scan_box = [&](int c) {
        // box c is wave-uniform.  Which lanes' own bound admits it?
        const float4 cmin = box_min[c], cmax = box_max[c];
        const float gx = gap(q.x, q.x, cmin.x, cmax.x), gy = gap(q.y, q.y, cmin.y, cmax.y),
                    gz = gap(q.z, q.z, cmin.z, cmax.z);
        const float lb = (gx * gx + gy * gy) + gz * gz;
        const bool need = live && lb <= best.wd;
        unsigned long long needing = __ballot(need);
        if (!needing) return;
        const int ci = c * kWave + lane;
        const int n = min(kWave, P - c * kWave);
        const float4 mine = ci < P ? sorted[ci] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        __syncthreads();
        s_cand[lane] = mine;
        __syncthreads();
        auto insert_passing = [&](unsigned long long pass) {   // re-checked: the worst entry shrinks as the list fills
            while (pass) {
                const int j = __builtin_ctzll(pass);
                pass &= pass - 1;
                const float4 p = s_cand[j];
                const int cid = (int)__float_as_uint(p.w);
                const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                if (best.better(d2, cid)) best.insert(d2, cid);
            }
        };
        if (__popcll(needing) <= kSparse) {
            // few queries want this box: lanes = its 64 points, one step per wanting query
            const int cid = (int)__float_as_uint(mine.w);
            while (needing) {
                const int ql = __builtin_ctzll(needing);
                needing &= needing - 1;
                const float qx = __shfl(q.x, ql), qy = __shfl(q.y, ql), qz = __shfl(q.z, ql);
                const float qwd = __shfl(best.wd, ql);
                const int qwid = __shfl(best.wid, ql), qid = __shfl(my_id, ql);
                const float dx = mine.x - qx, dy = mine.y - qy, dz = mine.z - qz;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                const unsigned long long pass =
                    __ballot(lane < n && cid != qid && (d2 < qwd || (d2 == qwd && cid < qwid)));
                if (lane == ql) insert_passing(pass);
            }
            return;
        }
        if (!need) return;
        // many queries want it: lanes = queries; a divergence-free pass records which points beat the worst entry
        unsigned long long pass = 0;
        for (int j = 0; j < n; j++) {
            const float4 p = s_cand[j];
            const int cid = (int)__float_as_uint(p.w);
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (cid != my_id && best.better(d2, cid)) pass |= 1ull << j;
        }
        insert_passing(pass);
    };

    // ---- seed: own box and its Morton neighbours, enough points to fill the list ----
    const int s = (K + kWave) / kWave;
    const int seed_lo = max(0, b - s), seed_hi = min(nb - 1, b + s);
    scan_box(b);
    for (int o = 1; o <= s; o++) {
        if (b + o <= seed_hi) scan_box(b + o);
        if (b - o >= seed_lo) scan_box(b - o);
    }

    // ---- sweep: 64 box tests per step, nearest chunks (in Morton order) first ----
    const int nchunks = (nb + kWave - 1) / kWave;
    const int cb = b / kWave;
    for (int t = 0; t <= 2 * nchunks; t++) {
        const int chunk = cb + ((t + 1) >> 1) * ((t & 1) ? 1 : -1);
        if (chunk < 0 || chunk >= nchunks) continue;
        const int c = chunk * kWave + lane;
        float lb = kFltMax;
        if (c < nb && (c < seed_lo || c > seed_hi)) {
            const float4 cmin = box_min[c], cmax = box_max[c];
            const float gx = gap(wmin.x, wmax.x, cmin.x, cmax.x), gy = gap(wmin.y, wmax.y, cmin.y, cmax.y),
                        gz = gap(wmin.z, wmax.z, cmin.z, cmax.z);
            lb = (gx * gx + gy * gy) + gz * gz;
        }
        float reject = wave_max(live ? best.wd : -1.f);
        unsigned long long todo = __ballot(lb <= reject && c < nb && (c < seed_lo || c > seed_hi));
        while (todo) {
            const int bit = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float lb_bit = __shfl(lb, bit);
            if (lb_bit > reject) continue;      // the bound shrank since the ballot
            scan_box(chunk * kWave + bit);
            reject = wave_max(live ? best.wd : -1.f);
        }
    }

    if (!live) return;
    // ---- output: selection sort of the column, ascending by (distance, index) ----
    float out_d[3] = {kFltMax, kFltMax, kFltMax};
    float last_d = -1.f;
    int last_id = -1;
    for (int r = 0; r < K; r++) {
        float bd = kFltMax;
        int bid = 0x7fffffff;
        for (int j = 0; j < K; j++) {
            const float v = best.d[j * kWave];
            const int vi = best.id[j * kWave];
            const bool after_last = v > last_d || (v == last_d && vi > last_id);
            if (after_last && (v < bd || (v == bd && vi < bid))) {
                bd = v;
                bid = vi;
            }
        }
        last_d = bd;
        last_id = bid;
        if (MEAN3) {
            if (r < 3) out_d[r] = bd;
        } else {
            dists[(size_t)my_id * K + r] = bd;
            indices[(size_t)my_id * K + r] = bid == 0x7fffffff ? -1 : bid;
        }
    }
    if (MEAN3) mean3[my_id] = ((out_d[0] + out_d[1]) + out_d[2]) / 3.0f;
}


// ---- distIndexQ: K nearest among a CANDIDATE subset for a QUERY subset (SimpleKNN::knn_indexQ, simple_knn.cu:523-660) ----
// The reference walks its Morton boxes from the query outward and filters by an is_neighbour mask.  This operator has no
// caller in the reference tree, so it is built for exactness and simplicity, not for the million-point case: one lane per
// query, the candidate list streamed through LDS in tiles that every lane scans with broadcast reads (O(Q * N) distance
// evaluations, ~8 flops each).  A query's K-best list lives in ITS output row, kept ascending by (distance, index); an
// insertion shifts the tail (rare once the list is warm).  Duplicate candidate indices count once (the reference's mask):
// the first kernel elects one owner per distinct index.  The point itself is excluded by index, as in the reference.
constexpr int kQTile = 1024;

__global__ __launch_bounds__(256) void knnq_owner_init_kernel(int P, int* __restrict__ owner)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) owner[i] = 0x7fffffff;
}

__global__ __launch_bounds__(256) void knnq_owner_kernel(int P, int N, const int* __restrict__ n_indices, int* __restrict__ owner)
{
    for (int j = blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) {
        const int n = n_indices[j];
        if (n >= 0 && n < P) atomicMin(&owner[n], j);   // lowest position wins: deterministic
    }
}

__global__ __launch_bounds__(256) void knnq_kernel(int P, int K, const float* __restrict__ pts, int Q,
                                                   const int* __restrict__ q_indices, int N, const int* __restrict__ n_indices,
                                                   const int* __restrict__ owner, float* __restrict__ dists,
                                                   int* __restrict__ indices)
{
    __shared__ float4 s_c[kQTile];
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < Q;
    int qi = live ? q_indices[q] : -1;
    const bool ok = qi >= 0 && qi < P;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (ok) {
        qx = pts[3 * qi];
        qy = pts[3 * qi + 1];
        qz = pts[3 * qi + 2];
    }
    float* const bd = dists + (size_t)(live ? q : 0) * K;
    int* const bi = indices + (size_t)(live ? q : 0) * K;
    if (live)
        for (int j = 0; j < K; j++) {
            bd[j] = kFltMax;
            bi[j] = -1;
        }
    int filled = 0;
    float wd = kFltMax;       // the list's last entry (worst kept): (wd, wi)
    int wi = 0x7fffffff;
    for (int base = 0; base < N; base += kQTile) {
        __syncthreads();
        for (int t = threadIdx.x; t < kQTile; t += 256) {
            const int j = base + t;
            float4 c = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (j < N) {
                const int n = n_indices[j];
                if (n >= 0 && n < P && owner[n] == j) c = make_float4(pts[3 * n], pts[3 * n + 1], pts[3 * n + 2], __int_as_float(n));
            }
            s_c[t] = c;
        }
        __syncthreads();
        if (!live || !ok) continue;
        const int cnt = min(kQTile, N - base);
        for (int t = 0; t < cnt; t++) {
            const float4 c = s_c[t];
            const int n = __float_as_int(c.w);
            if (n < 0 || n == qi) continue;
            const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (filled == K && !(d < wd || (d == wd && n < wi))) continue;
            // insert (d, n) into the ascending row; the tail moves one slot down, the last entry drops out when full
            int pos = filled < K ? filled : K - 1;
            while (pos > 0) {
                const float pd = bd[pos - 1];
                const int pi = bi[pos - 1];
                if (pd < d || (pd == d && pi < n)) break;
                bd[pos] = pd;
                bi[pos] = pi;
                pos--;
            }
            bd[pos] = d;
            bi[pos] = n;
            if (filled < K) filled++;
            if (filled == K) {
                wd = bd[K - 1];
                wi = bi[K - 1];
            }
        }
    }
}

}  // namespace

extern "C" {

int r3dgs_knn_max_k(void) { return kMaxK; }

size_t r3dgs_knn_workspace_bytes(int P)
{
    size_t out = 0;
    r3::guarded_call([&]() {
        if (P <= 0) return 0;
        const size_t temp = sort_temp_bytes((size_t)P);
        KnnWork w = KnnWork::carve(nullptr, (size_t)P, temp);
        out = (size_t)reinterpret_cast<uintptr_t>(w.temp) + temp + 256;
        return 0;
    });
    return out;
}

int r3dgs_knn(int P, int K, const float* points, float* dists, int* indices, float* mean_dist3, char* workspace,
              void* stream)
{
    return r3::guarded_call([&]() {
        if (P <= 0) return 0;
        if (mean_dist3) K = 3;
        if (K < 1 || K > kMaxK) throw r3::Error("K must be in [1, " + std::to_string(kMaxK) + "]");
        if (!points || !workspace || (!mean_dist3 && (!dists || !indices)))
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        size_t temp = sort_temp_bytes((size_t)P);
        KnnWork w = KnnWork::carve(workspace, (size_t)P, temp);
        const int nb = (P + kWave - 1) / kWave;
        const int g256 = (P + 255) / 256;
        bbox_init_kernel<<<1, 64, 0, s>>>(w.bbox);
        bbox_kernel<<<g256 < 256 ? g256 : 256, 256, 0, s>>>(P, points, w.bbox);
        morton_kernel<<<g256, 256, 0, s>>>(P, points, w.bbox, w.code, w.index);
        R3_HIP(rocprim::radix_sort_pairs(w.temp, temp, w.code, w.code_sorted, w.index, w.index_sorted, (size_t)P, 0, 30,
                                         s));
        gather_boxes_kernel<<<g256, 256, 0, s>>>(P, points, w.index_sorted, w.sorted, w.box_min, w.box_max);
        const size_t lds = kWave * sizeof(float4) + (size_t)K * kWave * 8;
        if (mean_dist3)
            knn_kernel<true><<<nb, kWave, lds, s>>>(P, K, w.sorted, w.box_min, w.box_max, nullptr, nullptr, mean_dist3);
        else
            knn_kernel<false><<<nb, kWave, lds, s>>>(P, K, w.sorted, w.box_min, w.box_max, dists, indices, nullptr);
        r3::check_launch("knn", s, false);
        return 0;
    });
}

size_t r3dgs_knn_query_workspace_bytes(int P) { return P > 0 ? 4 * (size_t)P + 256 : 256; }

int r3dgs_knn_query(int P, int K, const float* points, int Q, const int* q_indices, int N, const int* n_indices, float* dists,
                    int* indices, char* workspace, void* stream)
{
    return r3::guarded_call([&]() {
        if (Q <= 0) return 0;
        if (K < 1 || K > 4096) throw r3::Error("K must be in [1, 4096]");
        if (P < 0 || N < 0) throw r3::Error("negative size");
        if (!q_indices || !dists || !indices || !workspace || (P > 0 && !points) || (N > 0 && !n_indices))
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        int* owner = reinterpret_cast<int*>(workspace);
        if (P > 0) knnq_owner_init_kernel<<<std::min((P + 255) / 256, 1024), 256, 0, s>>>(P, owner);
        if (N > 0) knnq_owner_kernel<<<std::min((N + 255) / 256, 1024), 256, 0, s>>>(P, N, n_indices, owner);
        knnq_kernel<<<(Q + 255) / 256, 256, 0, s>>>(P, K, points, Q, q_indices, N, n_indices, owner, dists, indices);
        r3::check_launch("knn_query", s, false);
        return 0;
    });
}

}  // extern "C"
