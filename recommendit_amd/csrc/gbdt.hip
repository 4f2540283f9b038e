// LambdaMART forward (sum of reached leaf values) for gfx950 -- replaces lgb.Booster.predict
// behind LightGBMRanker.predict (reference src/models/ranker.py:161-174) and the text-model
// loader behind LightGBMRanker.load (ranker.py:212-226).
//
// Neither HBM- nor MFMA-bound: the forest (<1 MB) is cache resident and each candidate chases
// pointers.  Layout: the forest is cut into chunks of whole trees whose nodes fit LDS; a
// workgroup = 64 candidates (one per lane, features staged TRANSPOSED in LDS so a lane's
// feature fetch is bank = lane) x one tree chunk (nodes + leaves staged in LDS, 4 waves take
// every 4th tree).  grid = (candidate tiles) x (chunks), so one 500-candidate request still
// fills the chip.  Partial sums are combined in fixed order in double => deterministic.
//
// Decision rule restated from LightGBM (Tree::NumericalDecision / CategoricalDecision):
//   numerical: NaN -> 0 unless missing_type==NaN; "missing" (zero|nan by type) goes default
//   side; else left iff (double)fval <= threshold.  categorical: bitset membership.
#include "common.h"
#include "recommendit_hip.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

namespace {

struct Node {  // 32 bytes
  double thr;
  int left, right;  // >=0 internal node (tree-local), <0 => ~leaf (tree-local)
  int feat;
  int dtype;        // LightGBM decision_type byte
  int pad0, pad1;
};

// Compact node: numerical split only.  x <= thr (double) for an f32 x is the same as x <= thr32 with thr32 the largest
// f32 not above thr, so the comparison is exact in f32; children are tree-local int8 (>= 0 internal, < 0 => ~leaf).
struct Node8 {
  float thr;
  int8_t left, right;
  uint8_t feat;
  uint8_t flags;  // bit0 default_left, bits1-2 missing type (0 none, 1 zero, 2 NaN)
};
static_assert(sizeof(Node8) == 8, "Node8 must be 8 bytes");
constexpr int N8_CAP = 2048;   // compact nodes per chunk (16 KB of LDS)
constexpr int L8_CAP = 2112;   // leaves per chunk (16.5 KB)
constexpr int T8_CAP = 32;     // trees per chunk (8 per wave)

constexpr int NODE_CAP = 1280;  // nodes per chunk staged in LDS (40 KB)
constexpr int LEAF_CAP = 1536;  // leaves per chunk staged in LDS (12 KB)
constexpr int F_MAX = 128;      // features staged per candidate tile (64 x F x 4 B <= 32 KB)

struct Forest {
  int n_trees = 0, n_features = 0;
  std::vector<std::string> feature_names;
  std::vector<double> imp_split, imp_gain;
  bool average_output = false;
  // host copies
  std::vector<Node> nodes;
  std::vector<double> leaves;
  std::vector<int> tree_node_off, tree_leaf_off, tree_root;  // per tree (+1 sentinel for offs)
  std::vector<int> chunk_tree_start;                         // [n_chunks+1]
  std::vector<int> cat_boundaries, cat_words, tree_cat_b_off, tree_cat_w_off;
  // device
  Node* d_nodes = nullptr; double* d_leaves = nullptr;
  int *d_tree_node_off = nullptr, *d_tree_leaf_off = nullptr, *d_tree_root = nullptr, *d_chunk = nullptr;
  int *d_cat_b = nullptr, *d_cat_w = nullptr, *d_tree_cat_b_off = nullptr, *d_tree_cat_w_off = nullptr;
  double* d_part = nullptr; int64_t part_elems = 0;
  // compact form (all-numerical forests whose trees have <= 127 internal nodes): 8-byte nodes, f32 thresholds
  bool compact = false;
  std::vector<Node8> nodes8;
  std::vector<int> chunk8;  // [n_chunks8+1] tree starts
  Node8* d_nodes8 = nullptr; int* d_chunk8 = nullptr;
  std::vector<int> depth8;   // per tree: longest root-to-leaf path (internal nodes visited)
  int* d_depth8 = nullptr;
  bool simple8 = false;      // every split has missing type None: NaN features are cleaned to 0 once per tile
  float zero_thr32 = 0.f;
  // walk-ordered records (simple8 forests): see gbdt_walk_kernel
  bool bfs = false;
  std::vector<uint2> rec;            // per chunk: 8-byte records, chunk c at rec_chunk_off[c]
  std::vector<int> rec_chunk_off;    // [n_chunks8+1] record index of each chunk's first record
  std::vector<int> rec_root;         // per tree: byte offset of its root record inside its chunk
  uint2* d_rec = nullptr; int* d_rec_chunk_off = nullptr; int* d_rec_root = nullptr;
};

struct PredArgs {
  const Node* nodes; const double* leaves;
  const int *tree_node_off, *tree_leaf_off, *tree_root, *chunk;
  const int *cat_b, *cat_w, *tree_cat_b_off, *tree_cat_w_off;
  const float* X; int64_t n; int F; int ldx;
  double* part;  // [n_chunks, n]
};

__device__ __forceinline__ bool decide_left(float fv, const Node& nd, const int* cat_b, const int* cat_w) {
  double f = (double)fv;
  const int dt = nd.dtype;
  if (dt & 1) {  // categorical
    if (isnan(f)) return false;
    const int iv = (int)f;
    if (iv < 0) return false;
    const int ci = (int)nd.thr;
    const int b0 = cat_b[ci], b1 = cat_b[ci + 1];
    const int wi = iv >> 5;
    if (wi >= b1 - b0) return false;
    return (cat_w[b0 + wi] >> (iv & 31)) & 1;
  }
  const int missing = (dt >> 2) & 3;
  const bool is_nan = isnan(f);
  if (is_nan && missing != 2) f = 0.0;
  const bool is_missing = (missing == 1) ? (fabs(f) <= 1e-35) : ((missing == 2) ? is_nan : false);
  if (is_missing) return (dt & 2) != 0;
  return f <= nd.thr;
}

__global__ __launch_bounds__(256) void gbdt_predict_kernel(PredArgs a) {
  __shared__ __attribute__((aligned(16))) Node nS[NODE_CAP];
  __shared__ double lS[LEAF_CAP];
  __shared__ float xS[F_MAX * 64];  // transposed: xS[f*64 + lane]
  __shared__ double red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chunk = blockIdx.y;
  const int t0 = a.chunk[chunk], t1 = a.chunk[chunk + 1];
  const int n0 = a.tree_node_off[t0], n1 = a.tree_node_off[t1];
  const int l0 = a.tree_leaf_off[t0], l1 = a.tree_leaf_off[t1];
  const bool in_lds = (n1 - n0) <= NODE_CAP && (l1 - l0) <= LEAF_CAP;
  if (in_lds) {
    const int4* src = reinterpret_cast<const int4*>(a.nodes + n0);
    int4* dst = reinterpret_cast<int4*>(nS);
    for (int i = tid; i < (n1 - n0) * 2; i += 256) dst[i] = src[i];
    for (int i = tid; i < (l1 - l0); i += 256) lS[i] = a.leaves[l0 + i];
  }
  // the workgroup keeps its tree chunk in LDS and walks candidate tile after candidate tile through it: the forest is
  // read gridDim.x times in all instead of once per 64 candidates (1.4 GB of L2 traffic per 128k candidates before)
  const int64_t n_tiles = (a.n + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const int64_t row = tile * 64 + lane;
  __syncthreads();   // previous tile's walks and reduction are done with xS / red
  {  // lane = candidate, wave w takes features w, w+4, ...: conflict-free writes of the transposed tile
    const int64_t gr = tile * 64 + lane;
    const float* xr = a.X + (gr < a.n ? gr : 0) * a.ldx;
    for (int f = w; f < a.F; f += 4) xS[f * 64 + lane] = (gr < a.n) ? xr[f] : 0.f;
  }
  __syncthreads();
  double acc = 0.0;
  for (int t = t0 + w; t < t1; t += 4) {
    const int nb = a.tree_node_off[t], lb = a.tree_leaf_off[t];
    const Node* nodes = in_lds ? (nS + (nb - n0)) : (a.nodes + nb);
    const int* cb = a.cat_b ? a.cat_b + a.tree_cat_b_off[t] : nullptr;
    const int* cw = a.cat_w ? a.cat_w + a.tree_cat_w_off[t] : nullptr;
    int node = a.tree_root[t];
    while (node >= 0) {
      const Node nd = nodes[node];
      const float fv = xS[nd.feat * 64 + lane];
      node = decide_left(fv, nd, cb, cw) ? nd.left : nd.right;
    }
    const int leaf = ~node;
    acc += in_lds ? lS[lb - l0 + leaf] : a.leaves[lb + leaf];
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && row < a.n) a.part[(size_t)chunk * a.n + row] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  }
}

// Compact-forest traversal: 8-byte nodes, exact f32 comparisons, dynamic LDS sized to the forest chunk and the feature
// count (47 KB at 50 features => 3 workgroups per CU instead of 1: the walk is a chain of dependent LDS reads, so
// resident waves are what hides its latency).  Same per-wave tree order and same reduction as the general kernel.
struct Pred8Args {
  const Node8* nodes; const double* leaves;
  const int *tree_node_off, *tree_leaf_off, *tree_root, *chunk, *tree_depth;
  const float* X; int64_t n; int F; int ldx;
  float zero_thr;
  double* part;
};
template <bool SIMPLE>
__global__ __launch_bounds__(256, 3) void gbdt_predict8_kernel(Pred8Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  Node8* nS = reinterpret_cast<Node8*>(smem8);                   // [N8_CAP]
  double* lS = reinterpret_cast<double*>(nS + N8_CAP);           // [L8_CAP]
  double* red = lS + L8_CAP;                                     // [4][64]
  float* xS = reinterpret_cast<float*>(red + 4 * 64);            // [F][64] transposed
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chunk = blockIdx.y;
  const int t0 = a.chunk[chunk], t1 = a.chunk[chunk + 1];
  const int n0 = a.tree_node_off[t0], n1 = a.tree_node_off[t1];
  const int l0 = a.tree_leaf_off[t0], l1 = a.tree_leaf_off[t1];
  {
    const uint2* src = reinterpret_cast<const uint2*>(a.nodes + n0);
    uint2* dst = reinterpret_cast<uint2*>(nS);
    for (int i = tid; i < (n1 - n0); i += 256) dst[i] = src[i];
    for (int i = tid; i < (l1 - l0); i += 256) lS[i] = a.leaves[l0 + i];
  }
  // the workgroup keeps its tree chunk in LDS and walks candidate tile after candidate tile through it: the forest is
  // read gridDim.x times in all instead of once per 64 candidates (1.4 GB of L2 traffic per 128k candidates before)
  const int64_t n_tiles = (a.n + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
  const int64_t row = tile * 64 + lane;
  __syncthreads();   // previous tile's walks and reduction are done with xS / red
  {  // lane = candidate, wave w takes features w, w+4, ...: conflict-free writes of the transposed tile
    const int64_t gr = tile * 64 + lane;
    const float* xr = a.X + (gr < a.n ? gr : 0) * a.ldx;
    for (int f = w; f < a.F; f += 4) {
      float x = (gr < a.n) ? xr[f] : 0.f;
      if (SIMPLE && x != x) x = 0.f;   // missing type None: a NaN feature decides like 0.0 at every node
      xS[f * 64 + lane] = x;
    }
  }
  __syncthreads();
  double acc = 0.0;
  // each lane walks IL trees at once: the walks are chains of two dependent LDS reads per level (node, then the
  // feature it names), so independent chains are what fills the LDS pipe; leaves are added in tree order (as before)
  constexpr int IL = 4;
  for (int tb = t0 + w; tb < t1; tb += 4 * IL) {
    const Node8* nodes[IL];
    int node[IL];
#pragma unroll
    for (int j = 0; j < IL; ++j) {
      const int t = tb + 4 * j;
      const bool on = t < t1;
      nodes[j] = nS + (a.tree_node_off[on ? t : tb] - n0);
      node[j] = on ? a.tree_root[t] : -1;
    }
    // The walk is issue-bound (PMC: the SIMDs issue 86 % of the time), so the loop carries no per-lane branch: every
    // lane takes `dmax` steps (the deepest of the IL trees); a lane that has reached a leaf (node < 0) keeps it.
    int dmax = 0;
#pragma unroll
    for (int j = 0; j < IL; ++j) {
      const int t = tb + 4 * j;
      const int dj = t < t1 ? a.tree_depth[t] : 0;
      dmax = dj > dmax ? dj : dmax;
    }
    for (int lvl = 0; lvl < dmax; ++lvl) {
#pragma unroll
      for (int j = 0; j < IL; ++j) {
        const int cur = node[j];
        const Node8 nd = nodes[j][cur > 0 ? cur : 0];
        const float x = xS[nd.feat * 64 + lane];
        bool left;
        if (SIMPLE) {
          left = x <= nd.thr;
        } else {
          const int m = (nd.flags >> 1) & 3;
          const bool is_nan = x != x;
          const float f = (is_nan && m != 2) ? 0.f : x;
          const bool miss = (m == 1) ? (fabsf(f) <= a.zero_thr) : (m == 2 ? is_nan : false);
          left = miss ? (nd.flags & 1) : (f <= nd.thr);
        }
        const int nxt = left ? (int)nd.left : (int)nd.right;
        node[j] = cur >= 0 ? nxt : cur;
      }
    }
#pragma unroll
    for (int j = 0; j < IL; ++j) {
      const int t = tb + 4 * j;
      if (t < t1) acc += lS[a.tree_leaf_off[t] - l0 + ~node[j]];
    }
  }
  red[w * 64 + lane] = acc;
  __syncthreads();
  if (w == 0 && row < a.n)
    a.part[(size_t)chunk * a.n + row] = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
  }
}

// ---- walk-ordered records: all-numerical forests without missing-value handling (missing type None everywhere) ----
// The walk is instruction-issue bound (rocprofv3 PMC: the SIMDs issue 86 % of the cycles), so the node format is built
// for the fewest instructions per level.  Per tree the nodes AND leaves are laid out breadth first as 8-byte records with
// the two children of a node adjacent (left at a 16-byte boundary, right 8 bytes later):
//   internal record  {f32 thr (rounded down), u16 byte offset of the LEFT child | (feature * 256) << 16}
//   leaf record      {+inf,                   u16 own byte offset             | (4 * tree-local leaf index) << 16}
// One level = one 8-byte LDS read, one 4-byte LDS read of the lane's feature, a compare and
// `next = (word1 & 0xFFFF) | (x > thr ? 8 : 0)`; a leaf record points at itself, so every lane simply takes
// depth(tree) steps (no per-lane branch, no exec masking) and then reads the leaf slot of where it stands.
constexpr int R_CAP = 4096;  // records per chunk (32 KB of LDS)
struct WalkArgs {
  const uint2* rec; const double* leaves;
  const int *tree_leaf_off, *chunk, *rec_chunk_off, *rec_root, *tree_depth;
  const float* X; int64_t n; int F; int ldx;
  double* part;
};
// 8 waves per workgroup share one staged forest chunk and one candidate tile (the kernel needs 32 registers: the 63 KB of
// LDS bound it to 2 workgroups per CU, i.e. 8 waves per CU with 4-wave workgroups -- too few for a walk that waits on
// LDS 60 % of the time)
constexpr int WALK_NW = 16;
__global__ __launch_bounds__(64 * WALK_NW, 2) void gbdt_walk_kernel(WalkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
  unsigned char* rS = smemw;                                             // [R_CAP] records
  double* lS = reinterpret_cast<double*>(rS + sizeof(uint2) * R_CAP);    // [L8_CAP]
  double* red = lS + L8_CAP;                                             // [WALK_NW][64]
  unsigned char* xS = reinterpret_cast<unsigned char*>(red + WALK_NW * 64);    // [F][64] floats, transposed
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chunk = blockIdx.y;
  const int t0 = a.chunk[chunk], t1 = a.chunk[chunk + 1];
  const int r0 = a.rec_chunk_off[chunk], r1 = a.rec_chunk_off[chunk + 1];
  const int l0 = a.tree_leaf_off[t0], l1 = a.tree_leaf_off[t1];
  for (int i = tid; i < (r1 - r0); i += 64 * WALK_NW) reinterpret_cast<uint2*>(rS)[i] = a.rec[r0 + i];
  for (int i = tid; i < (l1 - l0); i += 64 * WALK_NW) lS[i] = a.leaves[l0 + i];
  const unsigned char* xL = xS + lane * 4;
  const int64_t n_tiles = (a.n + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 64 + lane;
    __syncthreads();   // previous tile's walks and reduction are done with xS / red (first pass: records are staged)
    {  // lane = candidate, wave w takes features w, w+4, ...: the transposed tile xS[f][candidate] is written without
       // bank conflicts (lane-consecutive); the strided global reads of a candidate's row are absorbed by L1
      const int64_t gr = tile * 64 + lane;
      const float* xr = a.X + (gr < a.n ? gr : 0) * a.ldx;
      for (int f = w; f < a.F; f += WALK_NW) {
        float x = (gr < a.n) ? xr[f] : 0.f;
        if (x != x) x = 0.f;         // missing type None: a NaN feature decides like 0.0 at every node
        reinterpret_cast<float*>(xS)[f * 64 + lane] = x;
      }
    }
    __syncthreads();
    double acc = 0.0;
    constexpr int IL = 2;            // trees walked at once by every lane (independent LDS chains)
    for (int tb = t0 + w; tb < t1; tb += WALK_NW * IL) {
      unsigned cur[IL];
      int dmax = 0;
#pragma unroll
      for (int j = 0; j < IL; ++j) {
        const int t = tb + WALK_NW * j;
        const bool on = t < t1;
        cur[j] = (unsigned)a.rec_root[on ? t : tb];
        const int dj = on ? a.tree_depth[t] : 0;
        dmax = dj > dmax ? dj : dmax;
      }
      for (int lvl = 0; lvl < dmax; ++lvl) {
        bool moved = false;
#pragma unroll
        for (int j = 0; j < IL; ++j) {
          const uint2 r = *reinterpret_cast<const uint2*>(rS + cur[j]);
          const float x = *reinterpret_cast<const float*>(xL + (r.y >> 16));
          const unsigned nxt = (r.y & 0xFFFFu) | (x > __uint_as_float(r.x) ? 8u : 0u);
          moved |= nxt != cur[j];
          cur[j] = nxt;
        }
        if (!__any(moved)) break;   // wave-uniform: every lane of the wave stands on a leaf in all IL trees
      }
#pragma unroll
      for (int j = 0; j < IL; ++j) {   // leaves are added in tree order: bitwise reproducible
        const int t = tb + WALK_NW * j;
        if (t < t1) acc += lS[a.tree_leaf_off[t] - l0 + (*reinterpret_cast<const unsigned*>(rS + cur[j] + 4) >> 18)];
      }
    }
    red[w * 64 + lane] = acc;
    __syncthreads();
    if (w == 0 && row < a.n) {
      double s = red[lane];
#pragma unroll
      for (int k = 1; k < WALK_NW; ++k) s += red[k * 64 + lane];   // fixed order
      a.part[(size_t)chunk * a.n + row] = s;
    }
  }
}

__global__ void gbdt_reduce_kernel(const double* __restrict__ part, int n_chunks, int64_t n, double scale, double* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int c = 0; c < n_chunks; ++c) s += part[(size_t)c * n + i];
  out[i] = s * scale;
}

// ---------------------------------- text-model parser -------------------------------------------
bool split_kv(const std::string& ln, std::string* k, std::string* v) {
  const size_t p = ln.find('=');
  if (p == std::string::npos) return false;
  *k = ln.substr(0, p);
  *v = ln.substr(p + 1);
  return true;
}
template <typename T>
std::vector<T> parse_list(const std::string& s) {
  std::vector<T> out;
  std::istringstream is(s);
  std::string tok;
  while (is >> tok) {
    if (sizeof(T) == sizeof(double)) out.push_back((T)strtod(tok.c_str(), nullptr));
    else out.push_back((T)strtoll(tok.c_str(), nullptr, 10));
  }
  return out;
}

// largest float not above x
static float f32_floor(double x) {
  float f = (float)x;
  if ((double)f > x) f = nextafterf(f, -INFINITY);
  return f;
}

static void build_compact(Forest* F, bool any_cat) {
  F->compact = false;
  if (any_cat || F->n_trees == 0 || F->n_features > 255) return;
  for (int t = 0; t < F->n_trees; ++t) {
    const int tn = F->tree_node_off[t + 1] - F->tree_node_off[t], tl = F->tree_leaf_off[t + 1] - F->tree_leaf_off[t];
    if (tn > 127 || tl > 128 || tn > N8_CAP || tl > L8_CAP) return;
  }
  F->nodes8.resize(F->nodes.size());
  for (size_t i = 0; i < F->nodes.size(); ++i) {
    const Node& nd = F->nodes[i];
    if (nd.dtype & 1) return;
    if (nd.left > 127 || nd.left < -128 || nd.right > 127 || nd.right < -128 || nd.feat < 0 || nd.feat > 255) return;
    Node8 c;
    c.thr = f32_floor(nd.thr);
    c.left = (int8_t)nd.left; c.right = (int8_t)nd.right; c.feat = (uint8_t)nd.feat;
    c.flags = (uint8_t)(((nd.dtype >> 1) & 1) | (((nd.dtype >> 2) & 3) << 1));
    F->nodes8[i] = c;
  }
  F->simple8 = true;
  for (const Node8& c : F->nodes8) if ((c.flags >> 1) & 3) F->simple8 = false;
  F->depth8.assign(F->n_trees, 0);
  for (int t = 0; t < F->n_trees; ++t) {   // longest path, iteratively (trees are small)
    const Node* nd = F->nodes.data() + F->tree_node_off[t];
    std::vector<std::pair<int, int>> stack;
    if (F->tree_root[t] >= 0) stack.push_back({F->tree_root[t], 1});
    int best = 0;
    while (!stack.empty()) {
      const std::pair<int, int> cur = stack.back();
      stack.pop_back();
      best = cur.second > best ? cur.second : best;
      if (nd[cur.first].left >= 0) stack.push_back({nd[cur.first].left, cur.second + 1});
      if (nd[cur.first].right >= 0) stack.push_back({nd[cur.first].right, cur.second + 1});
    }
    F->depth8[t] = best;
  }
  F->zero_thr32 = f32_floor(1e-35);
  F->chunk8.clear();
  F->chunk8.push_back(0);
  int cn = 0, cl = 0, ct = 0;
  for (int t = 0; t < F->n_trees; ++t) {
    const int tn = F->tree_node_off[t + 1] - F->tree_node_off[t], tl = F->tree_leaf_off[t + 1] - F->tree_leaf_off[t];
    if (ct > 0 && (cn + tn > N8_CAP || cl + tl > L8_CAP || ct >= T8_CAP)) { F->chunk8.push_back(t); cn = cl = ct = 0; }
    cn += tn; cl += tl; ++ct;
  }
  F->chunk8.push_back(F->n_trees);
  F->compact = true;
  // ---- walk-ordered records (gbdt_walk_kernel)
  F->bfs = false;
  if (!F->simple8) return;
  F->rec.clear(); F->rec_chunk_off.assign(1, 0); F->rec_root.assign(F->n_trees, 0);
  float inf_f = INFINITY; uint32_t inf_bits; memcpy(&inf_bits, &inf_f, 4);
  for (size_t c = 0; c + 1 < F->chunk8.size(); ++c) {
    const int t0 = F->chunk8[c], t1 = F->chunk8[c + 1];
    std::vector<uint2> cr;   // this chunk's records
    for (int t = t0; t < t1; ++t) {
      const Node8* nd = F->nodes8.data() + F->tree_node_off[t];
      if (cr.size() % 2) cr.push_back(make_uint2(inf_bits, 0));   // roots and sibling pairs start at 16-byte boundaries
      const size_t root = cr.size();
      F->rec_root[t] = (int)(root * 8);
      cr.push_back(make_uint2(0, 0));
      cr.push_back(make_uint2(inf_bits, (unsigned)((root + 1) * 8)));   // pad: keeps the next pair 16-byte aligned
      std::vector<std::pair<int, size_t>> queue;   // (node >= 0 or ~leaf, record index), breadth first
      queue.push_back({F->tree_root[t], root});
      for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int node = queue[qi].first;
        const size_t at = queue[qi].second;
        if (node < 0) {
          // upper half = 4 * (tree-local leaf index): read as a feature offset by the walk, so it must stay a valid,
          // 4-byte aligned offset into the feature tile (the value read there is irrelevant: thr = +inf)
          const unsigned leaf4 = (unsigned)(~node) * 4u;
          if (leaf4 > 0xFFFCu || (~node) > 127) return;
          cr[at] = make_uint2(inf_bits, (unsigned)(at * 8) | (leaf4 << 16));
        } else {
          const size_t pair = cr.size();
          cr.push_back(make_uint2(0, 0)); cr.push_back(make_uint2(0, 0));
          uint32_t tb; memcpy(&tb, &nd[node].thr, 4);
          cr[at] = make_uint2(tb, (unsigned)(pair * 8) | ((unsigned)nd[node].feat * 256u) << 16);
          queue.push_back({(int)nd[node].left, pair});
          queue.push_back({(int)nd[node].right, pair + 1});
        }
      }
    }
    if (cr.size() > (size_t)R_CAP) return;   // offsets must fit 16 bits and the chunk 32 KB of LDS
    F->rec.insert(F->rec.end(), cr.begin(), cr.end());
    F->rec_chunk_off.push_back((int)F->rec.size());
  }
  F->bfs = true;
}

int parse_model(const std::string& text, Forest* F) {
  std::istringstream in(text);
  std::string ln, k, v;
  bool in_tree = false, done = false;
  struct RawTree {
    int num_leaves = 0, num_cat = 0;
    std::vector<long long> split_feature, decision_type, left, right, cat_b, cat_w;
    std::vector<double> threshold, leaf_value, split_gain;
  };
  std::vector<RawTree> trees;
  int max_feature_idx = -1;
  while (!done && std::getline(in, ln)) {
    while (!ln.empty() && (ln.back() == '\r' || ln.back() == ' ')) ln.pop_back();
    if (ln.rfind("Tree=", 0) == 0) { trees.emplace_back(); in_tree = true; continue; }
    if (ln == "end of trees") { done = true; break; }
    if (!in_tree && ln == "average_output") { F->average_output = true; continue; }
    if (!split_kv(ln, &k, &v)) continue;
    if (!in_tree) {
      if (k == "feature_names") { std::istringstream is(v); std::string t; while (is >> t) F->feature_names.push_back(t); }
      else if (k == "max_feature_idx") max_feature_idx = atoi(v.c_str());
      else if (k == "num_class") { if (atoi(v.c_str()) != 1) { rihip_set_error("gbdt: num_class=%s unsupported", v.c_str()); return RIHIP_ERR_SHAPE; } }
      else if (k == "num_tree_per_iteration") { if (atoi(v.c_str()) != 1) { rihip_set_error("gbdt: num_tree_per_iteration=%s unsupported", v.c_str()); return RIHIP_ERR_SHAPE; } }
    } else {
      RawTree& t = trees.back();
      if (k == "num_leaves") t.num_leaves = atoi(v.c_str());
      else if (k == "num_cat") t.num_cat = atoi(v.c_str());
      else if (k == "split_feature") t.split_feature = parse_list<long long>(v);
      else if (k == "split_gain") t.split_gain = parse_list<double>(v);
      else if (k == "threshold") t.threshold = parse_list<double>(v);
      else if (k == "decision_type") t.decision_type = parse_list<long long>(v);
      else if (k == "left_child") t.left = parse_list<long long>(v);
      else if (k == "right_child") t.right = parse_list<long long>(v);
      else if (k == "leaf_value") t.leaf_value = parse_list<double>(v);
      else if (k == "cat_boundaries") t.cat_b = parse_list<long long>(v);
      else if (k == "cat_threshold") t.cat_w = parse_list<long long>(v);
      else if (k == "is_linear") { if (atoi(v.c_str()) != 0) { rihip_set_error("gbdt: linear trees unsupported"); return RIHIP_ERR_SHAPE; } }
    }
  }
  if (trees.empty() && !done) { rihip_set_error("gbdt: no trees found (not a LightGBM text model?)"); return RIHIP_ERR_IO; }
  F->n_trees = (int)trees.size();
  F->n_features = max_feature_idx >= 0 ? max_feature_idx + 1 : (int)F->feature_names.size();
  if ((int)F->feature_names.size() < F->n_features)
    for (int i = (int)F->feature_names.size(); i < F->n_features; ++i) F->feature_names.push_back("Column_" + std::to_string(i));
  F->imp_split.assign(F->n_features, 0.0);
  F->imp_gain.assign(F->n_features, 0.0);
  F->tree_node_off.push_back(0); F->tree_leaf_off.push_back(0);
  F->tree_cat_b_off.clear(); F->tree_cat_w_off.clear();
  bool any_cat = false;
  for (auto& t : trees) {
    const int ni = t.num_leaves > 1 ? t.num_leaves - 1 : 0;
    if ((int)t.leaf_value.size() < (t.num_leaves > 0 ? t.num_leaves : 1) && !(t.num_leaves <= 1 && t.leaf_value.empty())) {
      rihip_set_error("gbdt: tree with %d leaves has %zu leaf values", t.num_leaves, t.leaf_value.size()); return RIHIP_ERR_IO;
    }
    if (ni > 0 && ((int)t.split_feature.size() < ni || (int)t.threshold.size() < ni || (int)t.decision_type.size() < ni ||
                   (int)t.left.size() < ni || (int)t.right.size() < ni)) {
      rihip_set_error("gbdt: truncated tree arrays"); return RIHIP_ERR_IO;
    }
    F->tree_cat_b_off.push_back((int)F->cat_boundaries.size());
    F->tree_cat_w_off.push_back((int)F->cat_words.size());
    for (auto x : t.cat_b) F->cat_boundaries.push_back((int)x);
    for (auto x : t.cat_w) F->cat_words.push_back((int)x);
    if (t.num_cat > 0) any_cat = true;
    F->tree_root.push_back(ni > 0 ? 0 : ~0);
    for (int i = 0; i < ni; ++i) {
      Node nd;
      memset(&nd, 0, sizeof(nd));
      nd.thr = t.threshold[i]; nd.left = (int)t.left[i]; nd.right = (int)t.right[i];
      nd.feat = (int)t.split_feature[i]; nd.dtype = (int)t.decision_type[i];
      if (nd.feat < 0 || nd.feat >= F->n_features) { rihip_set_error("gbdt: split_feature %d out of range", nd.feat); return RIHIP_ERR_IO; }
      if ((nd.left >= ni) || (nd.right >= ni) || (~nd.left >= t.num_leaves && nd.left < 0) || (~nd.right >= t.num_leaves && nd.right < 0)) {
        rihip_set_error("gbdt: child index out of range"); return RIHIP_ERR_IO;
      }
      F->nodes.push_back(nd);
      F->imp_split[nd.feat] += 1.0;
      if (i < (int)t.split_gain.size()) F->imp_gain[nd.feat] += t.split_gain[i];
    }
    if (t.leaf_value.empty()) F->leaves.push_back(0.0);
    else for (int i = 0; i < (t.num_leaves > 0 ? t.num_leaves : 1); ++i) F->leaves.push_back(t.leaf_value[i]);
    F->tree_node_off.push_back((int)F->nodes.size());
    F->tree_leaf_off.push_back((int)F->leaves.size());
  }
  if (!any_cat) { F->cat_boundaries.clear(); F->cat_words.clear(); }
  // chunks of whole trees that fit the LDS staging budget (an oversize tree gets its own chunk, read from L2)
  F->chunk_tree_start.push_back(0);
  int cn = 0, cl = 0, ct = 0;
  for (int t = 0; t < F->n_trees; ++t) {
    const int tn = F->tree_node_off[t + 1] - F->tree_node_off[t], tl = F->tree_leaf_off[t + 1] - F->tree_leaf_off[t];
    if (ct > 0 && (cn + tn > NODE_CAP || cl + tl > LEAF_CAP || ct >= 16)) { F->chunk_tree_start.push_back(t); cn = cl = ct = 0; }
    cn += tn; cl += tl; ++ct;
  }
  F->chunk_tree_start.push_back(F->n_trees);
  build_compact(F, any_cat);
  return RIHIP_OK;
}

template <typename T>
int upload(const std::vector<T>& v, T** d) {
  *d = nullptr;
  if (v.empty()) return RIHIP_OK;
  if (hipMalloc((void**)d, sizeof(T) * v.size()) != hipSuccess || hipMemcpy(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) {
    rihip_set_error("gbdt: device upload failed"); return RIHIP_ERR_HIP;
  }
  return RIHIP_OK;
}

void destroy_forest(Forest* F) {
  if (!F) return;
  rihip_bump_generation();
  hipFree(F->d_nodes); hipFree(F->d_leaves); hipFree(F->d_tree_node_off); hipFree(F->d_tree_leaf_off); hipFree(F->d_tree_root);
  hipFree(F->d_chunk); hipFree(F->d_cat_b); hipFree(F->d_cat_w); hipFree(F->d_tree_cat_b_off); hipFree(F->d_tree_cat_w_off);
  hipFree(F->d_part); hipFree(F->d_nodes8); hipFree(F->d_chunk8); hipFree(F->d_depth8); hipFree(F->d_rec); hipFree(F->d_rec_chunk_off); hipFree(F->d_rec_root);
  delete F;
}

}  // namespace

extern "C" int rihip_gbdt_create_from_text(const char* text, int64_t len, void** handle) {
  RIHIP_REQUIRE(text && len > 0 && handle, RIHIP_ERR_ARG, "gbdt_create_from_text: bad arguments");
  Forest* F = new Forest();
  int rc = parse_model(std::string(text, (size_t)len), F);
  if (rc == RIHIP_OK && F->n_features > F_MAX) { rihip_set_error("gbdt: %d features > %d supported", F->n_features, F_MAX); rc = RIHIP_ERR_SHAPE; }
  if (rc == RIHIP_OK && F->n_trees > 0) {
    std::vector<Node> nodes = F->nodes;
    if (nodes.empty()) { Node z; memset(&z, 0, sizeof(z)); nodes.push_back(z); }
    rc = upload(nodes, &F->d_nodes);
    if (!rc) rc = upload(F->leaves, &F->d_leaves);
    if (!rc) rc = upload(F->tree_node_off, &F->d_tree_node_off);
    if (!rc) rc = upload(F->tree_leaf_off, &F->d_tree_leaf_off);
    if (!rc) rc = upload(F->tree_root, &F->d_tree_root);
    if (!rc) rc = upload(F->chunk_tree_start, &F->d_chunk);
    if (!rc && F->compact) {
      rc = upload(F->nodes8, &F->d_nodes8);
      if (!rc) rc = upload(F->chunk8, &F->d_chunk8);
      if (!rc) rc = upload(F->depth8, &F->d_depth8);
      if (!rc && F->bfs) { rc = upload(F->rec, &F->d_rec); if (!rc) rc = upload(F->rec_chunk_off, &F->d_rec_chunk_off); if (!rc) rc = upload(F->rec_root, &F->d_rec_root); }
    }
    if (!rc && !F->cat_boundaries.empty()) {
      rc = upload(F->cat_boundaries, &F->d_cat_b);
      if (!rc) rc = upload(F->cat_words, &F->d_cat_w);
      if (!rc) rc = upload(F->tree_cat_b_off, &F->d_tree_cat_b_off);
      if (!rc) rc = upload(F->tree_cat_w_off, &F->d_tree_cat_w_off);
    }
  }
  if (rc != RIHIP_OK) { destroy_forest(F); return rc; }
  *handle = F;
  return RIHIP_OK;
}

extern "C" int rihip_gbdt_load_text(const char* path, void** handle) {
  RIHIP_REQUIRE(path && handle, RIHIP_ERR_ARG, "gbdt_load_text: bad arguments");
  std::ifstream f(path, std::ios::binary);
  RIHIP_REQUIRE(f.good(), RIHIP_ERR_IO, "gbdt_load_text: cannot open %s", path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string s = ss.str();
  RIHIP_REQUIRE(!s.empty(), RIHIP_ERR_IO, "gbdt_load_text: %s is empty", path);
  return rihip_gbdt_create_from_text(s.data(), (int64_t)s.size(), handle);
}

extern "C" int rihip_gbdt_destroy(void* handle) { destroy_forest((Forest*)handle); return RIHIP_OK; }
extern "C" int rihip_gbdt_num_trees(void* handle) { return handle ? ((Forest*)handle)->n_trees : 0; }
extern "C" int rihip_gbdt_num_features(void* handle) { return handle ? ((Forest*)handle)->n_features : 0; }

// feature names joined by '\n' into buf (returns needed length incl. NUL)
extern "C" int64_t rihip_gbdt_feature_names(void* handle, char* buf, int64_t buf_len) {
  Forest* F = (Forest*)handle;
  if (!F) return 0;
  std::string s;
  for (size_t i = 0; i < F->feature_names.size(); ++i) { if (i) s += '\n'; s += F->feature_names[i]; }
  if (buf && buf_len > 0) { const size_t n = s.size() < (size_t)buf_len - 1 ? s.size() : (size_t)buf_len - 1; memcpy(buf, s.data(), n); buf[n] = 0; }
  return (int64_t)s.size() + 1;
}

// importance_type: 0 = split count, 1 = total gain; out[n_features]
extern "C" int rihip_gbdt_feature_importance(void* handle, int importance_type, double* out) {
  Forest* F = (Forest*)handle;
  RIHIP_REQUIRE(F && out, RIHIP_ERR_ARG, "gbdt_feature_importance: bad arguments");
  const std::vector<double>& v = importance_type == 0 ? F->imp_split : F->imp_gain;
  for (int i = 0; i < F->n_features; ++i) out[i] = v[i];
  return RIHIP_OK;
}

// X: device f32 [n, ldx] (first n_features columns used); out: device f64 [n] raw scores
extern "C" int rihip_gbdt_predict(void* handle, const float* X, int64_t n, int ldx, double* out, void* stream) {
  Forest* F = (Forest*)handle;
  RIHIP_REQUIRE(F && X && out && n >= 0, RIHIP_ERR_ARG, "gbdt_predict: bad arguments");
  RIHIP_REQUIRE(ldx >= F->n_features, RIHIP_ERR_SHAPE, "gbdt_predict: %d feature columns given, model needs %d", ldx, F->n_features);
  if (n == 0) return RIHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (F->n_trees == 0) { RIHIP_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(double) * n, st)); return RIHIP_OK; }
  const bool compact = F->compact && F->d_nodes8 != nullptr;
  const int n_chunks = compact ? (int)F->chunk8.size() - 1 : (int)F->chunk_tree_start.size() - 1;
  if (F->part_elems < (int64_t)n_chunks * n) {
    if (F->d_part) { hipFree(F->d_part); rihip_bump_generation(); }   // graphs that captured the old buffer are stale
    F->d_part = nullptr; F->part_elems = 0;
    RIHIP_CHECK_HIP(hipMalloc((void**)&F->d_part, sizeof(double) * (size_t)n_chunks * n));
    F->part_elems = (int64_t)n_chunks * n;
  }
  if (compact && F->bfs && F->d_rec) {
    WalkArgs c;
    c.rec = F->d_rec; c.leaves = F->d_leaves; c.tree_leaf_off = F->d_tree_leaf_off; c.chunk = F->d_chunk8;
    c.rec_chunk_off = F->d_rec_chunk_off; c.rec_root = F->d_rec_root; c.tree_depth = F->d_depth8;
    c.X = X; c.n = n; c.F = F->n_features; c.ldx = ldx; c.part = F->d_part;
    const size_t lds = sizeof(uint2) * R_CAP + sizeof(double) * (L8_CAP + WALK_NW * 64) + sizeof(float) * 64 * (size_t)F->n_features;
    static bool granted_w = false;
    if (!granted_w) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gbdt_walk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(uint2) * R_CAP + sizeof(double) * (L8_CAP + WALK_NW * 64) + sizeof(float) * 64 * 255));
      granted_w = true;
    }
    const int64_t n_tiles8 = (n + 63) / 64;
    int64_t per_chunk = (2 * RIHIP_NCU + n_chunks - 1) / n_chunks;     // 2 resident workgroups per CU over all chunks
    if (per_chunk > n_tiles8) per_chunk = n_tiles8;
    hipLaunchKernelGGL(gbdt_walk_kernel, dim3((unsigned)per_chunk, n_chunks), dim3(64 * WALK_NW), lds, st, c);
    RIHIP_CHECK_LAUNCH();
    const double scale_w = F->average_output ? 1.0 / (double)F->n_trees : 1.0;
    hipLaunchKernelGGL(gbdt_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, F->d_part, n_chunks, n, scale_w, out);
    RIHIP_CHECK_LAUNCH();
    return RIHIP_OK;
  }
  if (compact) {
    Pred8Args c;
    c.nodes = F->d_nodes8; c.leaves = F->d_leaves; c.tree_node_off = F->d_tree_node_off; c.tree_leaf_off = F->d_tree_leaf_off;
    c.tree_root = F->d_tree_root; c.chunk = F->d_chunk8; c.tree_depth = F->d_depth8; c.X = X; c.n = n; c.F = F->n_features; c.ldx = ldx;
    c.zero_thr = F->zero_thr32; c.part = F->d_part;
    const size_t lds = sizeof(Node8) * N8_CAP + sizeof(double) * (L8_CAP + 4 * 64) + sizeof(float) * 64 * (size_t)F->n_features;
    static bool granted = false;
    const int lds_max = (int)(sizeof(Node8) * N8_CAP + sizeof(double) * (L8_CAP + 4 * 64) + sizeof(float) * 64 * 255);
    if (!granted) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gbdt_predict8_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gbdt_predict8_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
      granted = true;
    }
    const int64_t n_tiles8 = (n + 63) / 64;
    int64_t per_chunk = (3 * RIHIP_NCU + n_chunks - 1) / n_chunks;     // 3 resident workgroups per CU over all chunks
    if (per_chunk > n_tiles8) per_chunk = n_tiles8;
    if (F->simple8) hipLaunchKernelGGL(gbdt_predict8_kernel<true>, dim3((unsigned)per_chunk, n_chunks), dim3(256), lds, st, c);
    else hipLaunchKernelGGL(gbdt_predict8_kernel<false>, dim3((unsigned)per_chunk, n_chunks), dim3(256), lds, st, c);
    RIHIP_CHECK_LAUNCH();
    const double scale8 = F->average_output ? 1.0 / (double)F->n_trees : 1.0;
    hipLaunchKernelGGL(gbdt_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, F->d_part, n_chunks, n, scale8, out);
    RIHIP_CHECK_LAUNCH();
    return RIHIP_OK;
  }
  PredArgs a;
  a.nodes = F->d_nodes; a.leaves = F->d_leaves; a.tree_node_off = F->d_tree_node_off; a.tree_leaf_off = F->d_tree_leaf_off;
  a.tree_root = F->d_tree_root; a.chunk = F->d_chunk; a.cat_b = F->d_cat_b; a.cat_w = F->d_cat_w;
  a.tree_cat_b_off = F->d_tree_cat_b_off; a.tree_cat_w_off = F->d_tree_cat_w_off;
  a.X = X; a.n = n; a.F = F->n_features; a.ldx = ldx; a.part = F->d_part;
  hipLaunchKernelGGL(gbdt_predict_kernel, dim3((unsigned)((n + 63) / 64), n_chunks), dim3(256), 0, st, a);
  RIHIP_CHECK_LAUNCH();
  const double scale = F->average_output ? 1.0 / (double)F->n_trees : 1.0;
  hipLaunchKernelGGL(gbdt_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, F->d_part, n_chunks, n, scale, out);
  RIHIP_CHECK_LAUNCH();
  return RIHIP_OK;
}
