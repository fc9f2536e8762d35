/*
 * nxs_gpu_replay.hip -- k_replay: the reference's capped min-heap replayed exactly; doc-shard merge
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"

#ifdef NXS_STATS
/* diagnostic build only (make variant XFLAGS=-DNXS_STATS), read back with nxsgpu_debug_rstats() */
__device__ unsigned long long g_rstats[8];	/* k_replay: queries, then 10 ns ticks per phase, candidates, inserts */
extern "C" void
nxsgpu_debug_rstats(unsigned long long *out, int reset)
{
	unsigned long long z[8] = { 0 };
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rstats), sizeof(z));
	if (reset) {
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_rstats), z, sizeof(z));
	}
}
#endif
/* ------------------------------------------------------------------ */
/* k_replay: the reference's heap, replayed exactly                     */
/* ------------------------------------------------------------------ */

/* heap_remove_min: src/algo/heap.c:133-189 (comparator: score only) */
__device__ static void
heap_remove_min(float *hs, uint32_t *hd, uint32_t *nitems, float *os, uint32_t *od)
{
	uint32_t i = 0, max_, left;

	*os = hs[0];
	*od = hd[0];
	if ((max_ = --(*nitems)) == 0) {
		return;
	}
	hs[0] = hs[max_];
	hd[0] = hd[max_];
	while ((left = i * 2 + 1) < max_) {
		const float ps = hs[i];
		const uint32_t pd = hd[i];
		const uint32_t right = i * 2 + 2;
		uint32_t smallest = i;

		if (hs[left] < ps) {
			smallest = left;
		}
		if (right < max_ && hs[right] < hs[smallest]) {
			smallest = right;
		}
		if (smallest == i) {
			break;
		}
		hs[i] = hs[smallest];
		hd[i] = hd[smallest];
		hs[smallest] = ps;
		hd[smallest] = pd;
		i = smallest;
	}
}

/* heap_add: src/algo/heap.c:58-124; caller has checked acceptance */
__device__ static void
heap_add(float *hs, uint32_t *hd, uint32_t *nitems, uint32_t cap, float s, uint32_t d)
{
	uint32_t i;

	if (*nitems == cap) {
		float ts; uint32_t td;
		heap_remove_min(hs, hd, nitems, &ts, &td);
	}
	i = (*nitems)++;
	hs[i] = s;
	hd[i] = d;
	while (i) {
		const uint32_t parent = (i - 1) / 2;
		const float ps = hs[parent];
		const uint32_t pd = hd[parent];
		if (s >= ps) {		/* heap.c:103 */
			break;
		}
		hs[parent] = s;
		hd[parent] = d;
		hs[i] = ps;
		hd[i] = pd;
		i = parent;
	}
}

/*
 * The same heap with its array ACROSS THE LANES of the wavefront (element i in
 * lane i, capacity <= 64): an element is read with v_readlane and written with
 * v_writelane, a few cycles each, where the LDS array cost a full LDS round
 * trip per access on lane 0 (about 1 us per heap_add: the replay of a single
 * query took 60-80 us, most of nxs_index_search()'s latency).  Every lane runs
 * the same scalar control flow; indices and values are wave-uniform.  Line for
 * line the functions above.
 */
__device__ __forceinline__ float
rh_gets(float hs, uint32_t i)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hs), (int)i));
}

__device__ __forceinline__ uint32_t
rh_getd(uint32_t hd, uint32_t i)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)hd, (int)i);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"	/* (M0 is reserved: nothing in these kernels uses it) */
__device__ __forceinline__ void
rh_set(float &hs, uint32_t &hd, uint32_t i, float s, uint32_t d)
{
	/* (no writelane builtin in this compiler; value and lane select are SGPRs) */
	const int sv = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s));
	const int dv = __builtin_amdgcn_readfirstlane((int)d);
	const int li = __builtin_amdgcn_readfirstlane((int)i);
	/* (one SGPR per VOP3 instruction on this target: the lane select goes through M0) */
	/* (s_nop: inline asm is outside the compiler's hazard recogniser; a scalar write
	 * of the lane select right in front of its vector use costs one idle cycle) */
	asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
	    : "+v"(hs), "+v"(hd) : "s"(sv), "s"(li), "s"(dv) : "m0");
}
#pragma clang diagnostic pop

/* heap_remove_min: src/algo/heap.c:133-189 */
__device__ static inline void
rheap_remove_min(float &hs, uint32_t &hd, uint32_t &nitems, float &os, uint32_t &od)
{
	uint32_t i = 0, max_, left;

	os = rh_gets(hs, 0);
	od = rh_getd(hd, 0);
	if ((max_ = --nitems) == 0) {
		return;
	}
	rh_set(hs, hd, 0, rh_gets(hs, max_), rh_getd(hd, max_));
	while ((left = i * 2 + 1) < max_) {
		const float ps = rh_gets(hs, i);
		const uint32_t pd = rh_getd(hd, i);
		const uint32_t right = i * 2 + 2;
		uint32_t smallest = i;

		if (rh_gets(hs, left) < ps) {
			smallest = left;
		}
		if (right < max_ && rh_gets(hs, right) < rh_gets(hs, smallest)) {
			smallest = right;
		}
		if (smallest == i) {
			break;
		}
		rh_set(hs, hd, i, rh_gets(hs, smallest), rh_getd(hd, smallest));
		rh_set(hs, hd, smallest, ps, pd);
		i = smallest;
	}
}

/* heap_add: src/algo/heap.c:58-124; caller has checked acceptance */
__device__ static inline void
rheap_add(float &hs, uint32_t &hd, uint32_t &nitems, uint32_t cap, float s, uint32_t d)
{
	uint32_t i;

	if (nitems == cap) {
		float ts; uint32_t td;
		rheap_remove_min(hs, hd, nitems, ts, td);
	}
	i = nitems++;
	rh_set(hs, hd, i, s, d);
	while (i) {
		const uint32_t parent = (i - 1) / 2;
		const float ps = rh_gets(hs, parent);
		const uint32_t pd = rh_getd(hd, parent);
		if (s >= ps) {		/* heap.c:103 */
			break;
		}
		rh_set(hs, hd, parent, s, d);
		rh_set(hs, hd, i, ps, pd);
		i = parent;
	}
}

/*
 * The lane-resident heap, worked on by ALL lanes at once (the form k_replay_coop
 * uses for the large heaps, below; there is a single 64-node block here and no
 * LDS).  Walking the levels on the scalar unit costs ~30 instructions per level
 * at one instruction per four cycles -- 0.5 us per heap_add measured, 30-50 us
 * of a single-term query's ~90 us.  Here every lane fetches its two children
 * with ds_bpermute, picks the smaller one (left on ties, heap.c:162-171) and says
 * whether the sinking element would move on from it; the two wave masks fix the
 * element's whole way down (lane b lies on it iff every ancestor sinks and points
 * towards b).  A rise fetches every node's parent the same way: the nodes
 * between the new leaf and the first ancestor that is <= the item (heap.c:103)
 * take their parent's pair.  Same comparisons, same final array.
 */
struct lph_lane_t {
	uint32_t	lvl;	/* depth of node `lane` */
	uint64_t	anc;	/* its proper ancestors, as a bit mask */
	uint64_t	dir;	/* ... and at which of them the way to it goes RIGHT */
};

__device__ __forceinline__ lph_lane_t
lph_lane_init(void)
{
	const uint32_t lane = threadIdx.x;
	lph_lane_t K;
	uint32_t a = lane;

	K.lvl = 31u - (uint32_t)__clz((int)(lane + 1));
	K.anc = K.dir = 0;
	while (a) {
		const uint32_t p = (a - 1) >> 1;
		K.anc |= 1ull << p;
		if (a == 2 * p + 2) {
			K.dir |= 1ull << p;
		}
		a = p;
	}
	return K;
}

/* the sinking loop of heap_remove_min (heap.c:149-187): element (es, ed) enters at
 * the root of a heap of n items; root_s = the new root's score */
__device__ __forceinline__ void
lph_sift_down(const lph_lane_t &K, float &hs, uint32_t &hd, uint32_t n, float es, uint32_t ed, float &root_s)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t cl = 2 * lane + 1, cr = cl + 1;
	/* (bpermute wraps the lane index: the loads of missing children are harmless) */
	const float ls_ = __shfl(hs, (int)(cl & 63)), rs_ = __shfl(hs, (int)(cr & 63));
	const uint32_t ld = (uint32_t)__shfl((int)hd, (int)(cl & 63)), rd = (uint32_t)__shfl((int)hd, (int)(cr & 63));
	const float ls = cl < n ? ls_ : INFINITY, rs = cr < n ? rs_ : INFINITY;
	const bool right = rs < ls;		/* heap.c:162-171: left unless the right child is strictly smaller */
	const float mcs = right ? rs : ls;
	const uint32_t mcd = right ? rd : ld;
	const uint64_t sink = ballot64(mcs < es);
	const uint64_t rmask = ballot64(right);
	const uint64_t bad = (~sink | (rmask ^ K.dir)) & K.anc;
	const uint64_t path = ballot64(bad == 0);
	const uint32_t c = 63u - (uint32_t)__builtin_clzll(path);	/* where the element comes to rest */

	if (lane_of(path & sink)) {
		hs = mcs;
		hd = mcd;
	}
	if (lane == c) {
		hs = es;
		hd = ed;
	}
	root_s = (sink & 1) ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mcs), 0)) : es;
}

/* heap_remove_min (heap.c:133-189); the caller has read the root */
__device__ __forceinline__ void
lph_remove_min(const lph_lane_t &K, float &hs, uint32_t &hd, uint32_t &n, float &root_s)
{
	if (--n == 0) {
		return;
	}
	lph_sift_down(K, hs, hd, n, rh_gets(hs, n), rh_getd(hd, n), root_s);	/* heap.c:146-147 */
}

/* heap_add (heap.c:58-124); the caller has checked acceptance (heap.c:68-74) */
__device__ __forceinline__ void
lph_add(const lph_lane_t &K, float &hs, uint32_t &hd, uint32_t &n, uint32_t cap, float s, uint32_t d, float &root_s)
{
	const uint32_t lane = threadIdx.x;

	if (n == cap) {
		lph_remove_min(K, hs, hd, n, root_s);
	}
	/* heap.c:96-122: the item enters at node i = n and rises past every ancestor
	 * that is larger, stopping at the first (from below) that is <= it */
	const uint32_t i = n++;
	const uint32_t par = (lane - 1) >> 1;		/* (lane 0: no parent; never used) */
	const float ps = __shfl(hs, (int)(par & 63));
	const uint32_t pd = (uint32_t)__shfl((int)hd, (int)(par & 63));
	const uint32_t ilo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)K.anc, (int)i);
	const uint32_t ihi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(K.anc >> 32), (int)i);
	const uint64_t anc_i = ((uint64_t)ihi << 32) | ilo;		/* proper ancestors of i */
	const uint64_t stop = anc_i & ~ballot64(s < hs);		/* ancestors that end the rise */
	/* the item lands on the chain node just below the nearest such ancestor (the
	 * root if there is none); the chain nodes below its place take their parent's pair */
	const uint64_t chain = anc_i | (1ull << i);
	const uint64_t below = stop ? chain & ~((2ull << (63 - __builtin_clzll(stop))) - 1) : chain;
	const uint32_t q = (uint32_t)__builtin_ctzll(below);		/* the shallowest of them: the item's place */
	if (lane_of(below)) {
		hs = ps;
		hd = pd;
	}
	if (lane == q) {
		hs = s;
		hd = d;
	}
	if (q == 0) {
		root_s = s;
	}
}

/*
 * The same heap once more for 64 < k <= REPLAY_LDS_K (the API's default limit is
 * 1000), in dynamic LDS as (score, doc) PAIRS: both children of a node come with
 * one read, and the element on the move stays in registers ("hole" form of the
 * reference's swaps: the same comparisons, the same final array).  One level is
 * one dependent LDS read instead of half a dozen: a heap_add on the 1000-entry
 * heap took 2.4 us with separate score / doc arrays (global memory or LDS alike).
 */
__device__ static inline void
lheap_remove_min(uint2 *h, uint32_t &nitems, float &os, uint32_t &od)
{
	uint32_t i = 0, max_, left;

	os = __uint_as_float(h[0].x);
	od = h[0].y;
	if ((max_ = --nitems) == 0) {
		return;
	}
	const uint2 e = h[max_];			/* heap.c:146-147: the last item goes to the root ... */
	const float ps = __uint_as_float(e.x);
	while ((left = i * 2 + 1) < max_) {		/* ... and sinks (heap.c:149-187) */
		const uint32_t right = left + 1;
		const uint2 cl = h[left];
		const uint2 cr = h[right < max_ ? right : left];
		uint32_t smallest = i;
		float ss = ps;
		uint2 cs = e;

		if (__uint_as_float(cl.x) < ps) {
			smallest = left;
			ss = __uint_as_float(cl.x);
			cs = cl;
		}
		if (right < max_ && __uint_as_float(cr.x) < ss) {
			smallest = right;
			cs = cr;
		}
		if (smallest == i) {
			break;
		}
		h[i] = cs;
		i = smallest;
	}
	h[i] = e;
}

__device__ static inline void
lheap_add(uint2 *h, uint32_t &nitems, uint32_t cap, float s, uint32_t d)
{
	uint32_t i;

	if (nitems == cap) {
		float ts; uint32_t td;
		lheap_remove_min(h, nitems, ts, td);
	}
	i = nitems++;
	while (i) {					/* heap.c:96-122 */
		const uint32_t parent = (i - 1) / 2;
		const uint2 pe = h[parent];
		if (s >= __uint_as_float(pe.x)) {	/* heap.c:103 */
			break;
		}
		h[i] = pe;
		i = parent;
	}
	h[i] = make_uint2(__float_as_uint(s), d);
}

template <int HEAP>
__global__ void __launch_bounds__(WAVE)
k_replay(const replay_args_t A)
{
	constexpr bool LDS_HEAP = HEAP == HEAP_REG;	/* (historic name: the k <= 64 heap) */
	extern __shared__ uint2 dyn_heap[];		/* HEAP_LDS: [k] */
	__shared__ uint32_t s_n;	/* (global-memory heap only) */
	__shared__ float s_min;

	const unsigned lane = threadIdx.x;
#ifdef NXS_STATS
	const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long rt1 = rt0, rt2 = rt0, rt3 = rt0;
	unsigned long long n_ins = 0, n_cand = 0;
#define	RSTAT(x)	x
#else
#define	RSTAT(x)
#endif
	const uint32_t q = A.qlist ? A.qlist[blockIdx.x] : blockIdx.x;
	float *hs = NULL;
	uint32_t *hd = NULL, cap;
	/* LDS_HEAP (k <= 64): the heap lives in these two registers, element i in lane i */
	float rhs = 0.0f;
	uint32_t rhd = 0, rn = 0;
	float rmin = 0.0f;
	const lph_lane_t KL = lph_lane_init();
	(void)KL;

	if (A.skip && A.skip[q]) {
		/* the query overflowed its candidate segments: its record says so (the
		 * owner re-runs it on the exact path; with sharding every rank sees it) */
		if (A.rec_base && lane == 0) {
			((uint32_t *)(A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes))[1] = NXSGPU_REC_INEXACT;
		}
		return;
	}
	if (LDS_HEAP) {
		cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(A.k, (uint32_t)WAVE));
	} else if (HEAP == HEAP_LDS) {
		/* (no heap_off: the filter pass of a limit > 64, every heap has room for k) */
		cap = A.heap_off ? (uint32_t)min((uint64_t)A.k, A.heap_off[q + 1] - A.heap_off[q]) : A.k;
	} else {
		hs = A.gheap_s + A.heap_off[q];
		hd = A.gheap_d + A.heap_off[q];
		cap = (uint32_t)min((uint64_t)A.k, A.heap_off[q + 1] - A.heap_off[q]);
	}
	if (lane == 0) {
		s_n = 0;
		s_min = 0.0f;
	}
	__syncthreads();

	/* candidates: groups in descending doc range, each already descending.
	 * Segment counts are fetched 64 at a time (one lane each) and empty
	 * segments -- most of them, once thresholds have warmed up -- are skipped
	 * without a memory round trip. */
	const qmeta_t qm = A.qmeta[q];
	for (int g0 = (int)qm.n_groups; g0 > 0 && cap; g0 -= WAVE) {
		const int gi = g0 - 1 - (int)lane;
		uint32_t cnt_l = 0;
		uint64_t sb_l = 0;
		if (gi >= 0) {
			const uint64_t seg_l = (uint64_t)qm.seg_first + gi;
			if (A.seg_cap) {
				cnt_l = A.seg_count[seg_l];
				sb_l = seg_l * A.seg_cap;
			} else {
				sb_l = A.seg_off[seg_l];
				cnt_l = (uint32_t)(A.seg_off[seg_l + 1] - sb_l);
			}
		}
		/*
		 * The candidates of these 64 segments, in feed order (segment lane
		 * ascending = doc range descending, then position), are packed 64
		 * to a load by a prefix sum over the counts; RU chunks are in flight.
		 */
		uint32_t incl = cnt_l;
		for (int o = 1; o < WAVE; o <<= 1) {
			const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= (unsigned)o) {
				incl += v;
			}
		}
		const uint32_t total = (uint32_t)__shfl((int)incl, WAVE - 1);
		RSTAT(n_cand += total; if (rt1 == rt0) rt1 = __builtin_amdgcn_s_memrealtime();)
		constexpr int RU = 4;
		/* heap.c:68-74 on one round of RU x 64 candidates (valid: index < bound) */
		auto consume = [&](uint32_t c0, uint32_t bound, const float (&scv)[RU], const uint32_t (&dcv)[RU]) {
#pragma unroll
			for (int u = 0; u < RU; u++) {
				const bool valid = c0 + u * WAVE + lane < bound;
				const float sc = scv[u];
				const uint32_t dc = dcv[u];
				/* heap.c:68-74: when full, an item <= the root is dropped
				 * without touching the heap */
				if constexpr (LDS_HEAP) {
					uint64_t pend = ballot64(valid && (rn < cap || sc > rmin));
					while (pend) {
						const int L = __builtin_ctzll(pend);
						const float v = rh_gets(sc, (uint32_t)L);
						const uint32_t dv = rh_getd(dc, (uint32_t)L);
#ifdef NXS_OLD_REGHEAP
						rheap_add(rhs, rhd, rn, cap, v, dv);
#else
						lph_add(KL, rhs, rhd, rn, cap, v, dv, rmin);
#endif
						RSTAT(n_ins++;)
						rn = (uint32_t)__builtin_amdgcn_readfirstlane((int)rn);
						if (A.log_cnt && lane == 0) {
							const uint32_t row = A.log_slot ? A.log_slot[q] : q;
							const uint32_t nl = A.log_cnt[row];
							if (nl < A.log_cap) {
								A.log_ids[(uint64_t)row * A.log_cap + nl] = A.doc_ids[dv];
								A.log_sc[(uint64_t)row * A.log_cap + nl] = v;
							}
							A.log_cnt[row] = nl + 1;
						}
#ifdef NXS_OLD_REGHEAP
						rmin = rh_gets(rhs, 0);
#endif
						pend &= pend - 1;
						pend &= ballot64(valid && (rn < cap || sc > rmin));
					}
					continue;
				}
				uint32_t nn = s_n;
				float mn = s_min;
				uint64_t pend = ballot64(valid && (nn < cap || sc > mn));
				while (pend) {
					const int L = __ffsll((long long)pend) - 1;
					const float v = __shfl(sc, L);
					const uint32_t dv = (uint32_t)__shfl((int)dc, L);
					RSTAT(n_ins++;)
					if (lane == 0) {
						uint32_t cnt = s_n;
						if constexpr (HEAP == HEAP_LDS) {
							lheap_add(dyn_heap, cnt, cap, v, dv);
						} else {
							heap_add(hs, hd, &cnt, cap, v, dv);
						}
						if (A.log_cnt) {
							const uint32_t row = A.log_slot ? A.log_slot[q] : q;
							const uint32_t nl = A.log_cnt[row];
							if (nl < A.log_cap) {
								A.log_ids[(uint64_t)row * A.log_cap + nl] = A.doc_ids[dv];
								A.log_sc[(uint64_t)row * A.log_cap + nl] = v;
							}
							A.log_cnt[row] = nl + 1;
						}
						s_n = cnt;
						s_min = HEAP == HEAP_LDS ? __uint_as_float(dyn_heap[0].x) : hs[0];
					}
					__syncthreads();
					nn = s_n;
					mn = s_min;
					pend &= pend - 1;
					pend &= ballot64(valid && (nn < cap || sc > mn));
				}
			}
		};
		if constexpr (!LDS_HEAP) {
			/*
			 * The exact passes (limit > 64) stream EVERY match of the query through
			 * here, thousands per segment: segment by segment, plain coalesced loads.
			 * (Packing the candidates of 64 segments by a prefix sum -- below, what
			 * the top-k pass needs for its many near-empty segments -- costs a 6-step
			 * cross-lane search per 64 candidates: 28 of the 30 ms of a default-limit
			 * batch.)
			 */
			uint64_t nonempty = ballot64(cnt_l != 0);
			while (nonempty) {
				const int sl = __builtin_ctzll(nonempty);
				nonempty &= nonempty - 1;
				const uint32_t scnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, sl);
				const uint64_t ssb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(sb_l >> 32), sl) << 32) |
				    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sb_l, sl);
				for (uint32_t c0 = 0; c0 < scnt; c0 += WAVE * RU) {
					float scv[RU];
					uint32_t dcv[RU];
#pragma unroll
					for (int u = 0; u < RU; u++) {
						const uint32_t j = c0 + u * WAVE + lane;
						scv[u] = 0.0f;
						dcv[u] = 0;
						if (j < scnt) {
							const uint64_t at = ssb + j;
							scv[u] = A.cand_sc[at];
							dcv[u] = A.cand_doc ? A.cand_doc[at] : (uint32_t)at;
						}
					}
					consume(c0, scnt, scv, dcv);
				}
			}
			continue;
		}
		for (uint32_t c0 = 0; c0 < total; c0 += WAVE * RU) {
			float scv[RU];
			uint32_t dcv[RU];
#pragma unroll
			for (int u = 0; u < RU; u++) {
				const uint32_t j = c0 + u * WAVE + lane;
				scv[u] = 0.0f;
				dcv[u] = 0;
				/* segment lane sl = first lane with incl > j (binary search
				 * over the lanes' inclusive sums) */
				uint32_t sl = 0;
#pragma unroll
				for (int step = 32; step >= 1; step >>= 1) {
					const uint32_t probe = (uint32_t)__shfl((int)incl, (int)(sl + step - 1));
					if (probe <= j) {
						sl += step;
					}
				}
				sl = min(sl, (uint32_t)WAVE - 1);
				const uint32_t s_incl = (uint32_t)__shfl((int)incl, (int)sl);
				const uint32_t s_cnt = (uint32_t)__shfl((int)cnt_l, (int)sl);
				const uint64_t s_sb = ((uint64_t)(uint32_t)__shfl((int)(sb_l >> 32), (int)sl) << 32) |
				    (uint32_t)__shfl((int)(uint32_t)sb_l, (int)sl);
				if (j < total) {
					const uint64_t at = s_sb + (j - (s_incl - s_cnt));
					scv[u] = A.cand_sc[at];
					dcv[u] = A.cand_doc ? A.cand_doc[at] : (uint32_t)at;
				}
			}
			consume(c0, total, scv, dcv);
		}
	}
	__syncthreads();

	if constexpr (LDS_HEAP) {
		RSTAT(rt2 = __builtin_amdgcn_s_memrealtime();)
		/* heap_sort (heap.c:197-221): repeated remove-min, placed from the back */
		const uint32_t cnt = rn;
		uint32_t n = rn;
		while (n) {
			const uint32_t last = n - 1;
			float ms; uint32_t mdoc;
#ifdef NXS_OLD_REGHEAP
			rheap_remove_min(rhs, rhd, n, ms, mdoc);
			n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
			rh_set(rhs, rhd, last, ms, mdoc);
#else
			ms = rh_gets(rhs, 0);
			mdoc = rh_getd(rhd, 0);
			lph_remove_min(KL, rhs, rhd, n, rmin);
			if (lane == last) {
				rhs = ms;
				rhd = mdoc;
			}
#endif
		}
		/* lane i holds result i */
		if (A.rec_base) {
			/* u32 count | u32 flags | u64 ids[k] | f32 scores[k]  (nxs_gpu.h) */
			uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
			uint64_t *r_ids = (uint64_t *)(rec + 8);
			float *r_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
			RSTAT(rt3 = __builtin_amdgcn_s_memrealtime();)
			if (lane < cnt) {
				r_ids[lane] = A.doc_ids[rhd];
				r_sc[lane] = rhs;
			}
			if (lane == 0) {
				((uint32_t *)rec)[0] = cnt;
			}
#ifdef NXS_STATS
			if (lane == 0) {
				const unsigned long long rt4 = __builtin_amdgcn_s_memrealtime();
				atomicAdd(&g_rstats[0], 1ull);
				atomicAdd(&g_rstats[1], rt1 - rt0);
				atomicAdd(&g_rstats[2], rt2 - rt1);
				atomicAdd(&g_rstats[3], rt3 - rt2);
				atomicAdd(&g_rstats[4], rt4 - rt3);
				atomicAdd(&g_rstats[5], n_cand);
				atomicAdd(&g_rstats[6], n_ins);
			}
#endif
			return;
		}
		const uint64_t ob = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
		if (lane < cnt) {
			A.out_ids[ob + lane] = A.doc_ids[rhd];
			A.out_sc[ob + lane] = rhs;
		}
		if (lane == 0) {
			A.out_count[q] = cnt;
		}
		return;
	}
	RSTAT(const unsigned long long rt_heap = __builtin_amdgcn_s_memrealtime();)
	/* heap_sort (heap.c:197-221): repeated remove-min, placed from the back */
	const uint32_t cnt = s_n;
	if constexpr (HEAP == HEAP_LDS) {
		if (lane == 0) {
			uint32_t n = cnt;
			while (n) {
				const uint32_t last = n - 1;
				float ms; uint32_t mdoc;
				lheap_remove_min(dyn_heap, n, ms, mdoc);
				dyn_heap[last] = make_uint2(__float_as_uint(ms), mdoc);
			}
		}
		__syncthreads();
#ifdef NXS_STATS
		if (lane == 0) {
			const unsigned long long rt4 = __builtin_amdgcn_s_memrealtime();
			atomicAdd(&g_rstats[0], 1ull);
			atomicAdd(&g_rstats[1], rt1 - rt0);
			atomicAdd(&g_rstats[2], rt_heap - rt1);
			atomicAdd(&g_rstats[3], rt4 - rt_heap);
			atomicMax(&g_rstats[4], rt4 - rt0);
			atomicAdd(&g_rstats[5], n_cand);
			atomicAdd(&g_rstats[6], n_ins);
			atomicMax(&g_rstats[7], n_ins * 1000000ull + n_cand / 16);
		}
#endif
		if (A.rec_base) {
			uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
			uint64_t *r_ids = (uint64_t *)(rec + 8);
			float *r_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
			for (uint32_t i = lane; i < cnt; i += WAVE) {
				r_ids[i] = A.doc_ids[dyn_heap[i].y];
				r_sc[i] = __uint_as_float(dyn_heap[i].x);
			}
			if (lane == 0) {
				((uint32_t *)rec)[0] = cnt;
			}
			return;
		}
		const uint64_t ob2 = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
		for (uint32_t i = lane; i < cnt; i += WAVE) {
			A.out_ids[ob2 + i] = A.doc_ids[dyn_heap[i].y];
			A.out_sc[ob2 + i] = __uint_as_float(dyn_heap[i].x);
		}
		if (lane == 0) {
			A.out_count[q] = cnt;
		}
		return;
	}
	if (lane == 0) {
		uint32_t n = cnt;
		while (n) {
			const uint32_t last = n - 1;
			float ms; uint32_t mdoc;
			heap_remove_min(hs, hd, &n, &ms, &mdoc);
			hs[last] = ms;
			hd[last] = mdoc;
		}
	}
	__syncthreads();
	if (A.rec_base) {
		/* u32 count | u32 flags | u64 ids[k] | f32 scores[k]  (nxs_gpu.h) */
		uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
		uint64_t *r_ids = (uint64_t *)(rec + 8);
		float *r_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
		for (uint32_t i = lane; i < cnt; i += WAVE) {
			r_ids[i] = A.doc_ids[hd[i]];
			r_sc[i] = hs[i];
		}
		if (lane == 0) {
			((uint32_t *)rec)[0] = cnt;
		}
		return;
	}
	const uint64_t ob = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
	for (uint32_t i = lane; i < cnt; i += WAVE) {
		A.out_ids[ob + i] = A.doc_ids[hd[i]];
		A.out_sc[ob + i] = hs[i];
	}
	if (lane == 0) {
		A.out_count[q] = cnt;
	}
}

/* ------------------------------------------------------------------ */
/* k_replay_coop: the same heap for 64 < k <= REPLAY_LDS_K, all 64 lanes  */
/* ------------------------------------------------------------------ */

/*
 * The API's default limit is 1000 (nxs_impl.h:39): a five-term OR over 10M docs
 * puts 3 600 items through the 1000-entry heap on average, 10 000 for the
 * heaviest query of a C3 batch -- one after the other (every heap_add sees the
 * array the previous one left: heap.c:58-124, and the final order among equal
 * scores is whatever those arrays make it, Q12).  On one lane an insertion is a
 * chain of ~13 dependent LDS round trips, or -- with the top of the heap in
 * registers -- ~25 scalar instructions per level at one instruction per four
 * cycles (a single wavefront's issue rate): 1.3-1.5 us either way, and the
 * heaviest query's replay (17-19 ms) set the pace of the whole batch.
 *
 * Same array (in LDS, (score, doc) pairs), same comparisons, same stores -- but
 * the wavefront works on one insertion TOGETHER, six levels at a time:
 *  - a sift-down looks at the 63-node subtree under its hole: lane b stands for
 *    descendant b (breadth-first) and loads that node's two children; each lane
 *    picks its smaller child (left on ties, heap.c:162-171) and says whether the
 *    sinking element would move on from it (child < element).  Those two wave
 *    masks fix the element's whole way through the six levels: lane b lies on it
 *    iff every ancestor sinks and points towards b -- two 64-bit constants per
 *    lane, a handful of vector instructions, no level-by-level walk.  The lanes on
 *    the way store their child's pair into their own node (one LDS instruction for
 *    all of them); the deepest one either holds the element's final place or
 *    names the root of the next six levels.  k = 1000 is ten levels: two rounds;
 *  - a sift-up loads all ancestors of the new leaf at once (lane t: t + 1 levels
 *    up); the first one that is <= the item (heap.c:103) ends the rise, the ones
 *    below it move down one level with one store;
 *  - every branch is wave-uniform; nothing is handed through LDS flags, no barrier.
 * About 0.4 us per insertion instead of 1.3-1.5.
 */
struct chp_lane_t {
	uint32_t	lvl;	/* depth of local node `lane` in a 63-node block (lane 63: unused) */
	uint64_t	anc;	/* its proper ancestors (local indices), as a bit mask */
	uint64_t	dir;	/* ... and at which of them the way to it goes RIGHT */
	bool		act;
};

__device__ __forceinline__ chp_lane_t
chp_lane_init(void)
{
	const uint32_t lane = threadIdx.x;
	chp_lane_t K;
	uint32_t a = lane;

	K.act = lane < 63;
	K.lvl = 31u - (uint32_t)__clz((int)(lane + 1));
	K.anc = K.dir = 0;
	while (a) {
		const uint32_t p = (a - 1) >> 1;
		K.anc |= 1ull << p;
		if (a == 2 * p + 2) {
			K.dir |= 1ull << p;
		}
		a = p;
	}
	if (!K.act) {
		K.anc = K.dir = 0;
	}
	return K;
}

/*
 * Six levels of heap_remove_min's sinking loop (heap.c:149-187) at once: the
 * element (score es) has its hole at node g; n = items in the heap.  The nodes it
 * passes take their smaller child's pair.  Returns true if it leaves the block
 * (g_out = the child it moves into: the next block's root), false if it comes to
 * rest (g_out = its place; the caller stores it).  top_s = what node g holds now
 * if the element moved on from it (the caller tracks the heap's minimum with it).
 */
__device__ __forceinline__ bool
chp_block(uint2 *h, const chp_lane_t &K, uint32_t g, uint32_t n, float es, uint32_t &g_out, float &top_s)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t G = (g << K.lvl) + lane;		/* this lane's node */
	const uint32_t cl = 2 * G + 1;
	float ls = INFINITY, rs = INFINITY;		/* a missing child never is the smaller one */
	uint32_t ld = 0, rd = 0;

	if (K.act && cl < n) {
		const uint2 a = h[cl];
		ls = __uint_as_float(a.x);
		ld = a.y;
		if (cl + 1 < n) {
			const uint2 b = h[cl + 1];
			rs = __uint_as_float(b.x);
			rd = b.y;
		}
	}
	const bool right = rs < ls;			/* heap.c:162-171: left unless the right child is strictly smaller */
	const float mcs = right ? rs : ls;
	const uint32_t mcd = right ? rd : ld;
	const uint64_t sink = ballot64(mcs < es);	/* the element would move on from these nodes */
	const uint64_t rmask = ballot64(right);
	/* on the element's way: every ancestor sinks and points here */
	const uint64_t bad = (~sink | (rmask ^ K.dir)) & K.anc;
	const uint64_t path = ballot64(K.act && bad == 0);	/* (never empty: the block's root has no ancestors) */
	const uint32_t c = 63u - (uint32_t)__builtin_clzll(path);	/* the deepest node on it */

	if (lane_of(path & sink)) {
		h[G] = make_uint2(__float_as_uint(mcs), mcd);
	}
	top_s = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mcs), 0));
	if ((sink >> c) & 1) {
		g_out = (uint32_t)__builtin_amdgcn_readlane((int)(cl + (right ? 1u : 0u)), (int)c);
		return true;
	}
	g_out = (uint32_t)__builtin_amdgcn_readlane((int)G, (int)c);
	return false;
}

/* heap_remove_min (heap.c:133-189) without the removed item (the caller has it:
 * the root); root_s = the new root's score */
__device__ __forceinline__ void
chp_remove_min(uint2 *h, const chp_lane_t &K, uint32_t &n, float &root_s)
{
	if (--n == 0) {
		return;
	}
	const uint2 e = h[n];				/* heap.c:146-147: the last item goes to the root ... */
	const float es = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)e.x));
	const uint32_t ed = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
	uint32_t g = 0, g_out;
	float top_s, dummy;
	bool more = chp_block(h, K, 0, n, es, g_out, top_s);	/* ... and sinks (heap.c:149-187) */
	root_s = (more || g_out != 0) ? top_s : es;
	while (more) {
		g = g_out;
		more = chp_block(h, K, g, n, es, g_out, dummy);
	}
	h[g_out] = make_uint2(__float_as_uint(es), ed);	/* (every lane: same address, same value) */
	asm volatile("" ::: "memory");
}

/* the rising loop of heap_add (heap.c:96-122): item (s, d) enters at index i */
__device__ __forceinline__ void
chp_sift_up(uint2 *h, uint32_t i, float s, uint32_t d, float &root_s)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t ip1 = i + 1;
	const uint32_t depth = 31u - (uint32_t)__clz((int)ip1);	/* proper ancestors of i */
	const bool has = lane < depth;				/* lane t: the ancestor t + 1 levels up */
	uint2 av = make_uint2(0, 0);

	if (has) {
		av = h[(ip1 >> (lane + 1)) - 1];
	}
	/* heap.c:103: the first ancestor that is <= the item ends the rise */
	const uint64_t rise = ballot64(has && s < __uint_as_float(av.x));
	const uint32_t u = (uint32_t)__builtin_ctzll(~rise);
	if (lane < u) {
		h[(ip1 >> lane) - 1] = av;			/* the ancestors below it move down one level */
	}
	const uint32_t pos = (ip1 >> u) - 1;
	h[pos] = make_uint2(__float_as_uint(s), d);
	if (pos == 0) {
		root_s = s;
	}
	asm volatile("" ::: "memory");
}

/* heap_add (heap.c:58-124); the caller has checked acceptance (heap.c:68-74) */
__device__ __forceinline__ void
chp_add(uint2 *h, const chp_lane_t &K, uint32_t &n, uint32_t cap, float s, uint32_t d, float &root_s)
{
	if (n == cap) {
		chp_remove_min(h, K, n, root_s);
	}
	chp_sift_up(h, n++, s, d, root_s);
}

__global__ void __launch_bounds__(WAVE)
k_replay_coop(const replay_args_t A)
{
	extern __shared__ uint2 coop_heap[];	/* [k] */
	const unsigned lane = threadIdx.x;
#ifdef NXS_STATS
	const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
	unsigned long long rt1 = rt0, n_ins = 0, n_cand = 0;
#endif
	const uint32_t q = A.qlist ? A.qlist[blockIdx.x] : blockIdx.x;
	uint2 *h = coop_heap;
	const chp_lane_t K = chp_lane_init();
	uint32_t n = 0;
	float rmin = 0.0f;	/* the root's score */

	if (A.skip && A.skip[q]) {
		if (A.rec_base && lane == 0) {
			((uint32_t *)(A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes))[1] = NXSGPU_REC_INEXACT;
		}
		return;
	}
	/* (no heap_off: the filter pass of a limit > 64, every heap has room for k) */
	const uint32_t cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)(A.heap_off ?
	    (uint32_t)min((uint64_t)A.k, A.heap_off[q + 1] - A.heap_off[q]) : A.k));
	const qmeta_t qm = A.qmeta[q];
	const uint32_t log_row = A.log_cnt ? (A.log_slot ? A.log_slot[q] : q) : 0;

	/* candidates: ranges in descending doc order, each already descending; segment
	 * by segment, plain coalesced loads, RU x 64 in flight */
	for (int g0 = (int)qm.n_groups; g0 > 0 && cap; g0 -= WAVE) {
		const int gi = g0 - 1 - (int)lane;
		uint32_t cnt_l = 0;
		uint64_t sb_l = 0;
		if (gi >= 0) {
			const uint64_t seg_l = (uint64_t)qm.seg_first + gi;
			if (A.seg_cap) {
				cnt_l = A.seg_count[seg_l];
				sb_l = seg_l * A.seg_cap;
			} else {
				sb_l = A.seg_off[seg_l];
				cnt_l = (uint32_t)(A.seg_off[seg_l + 1] - sb_l);
			}
		}
		RSTAT(if (rt1 == rt0) rt1 = __builtin_amdgcn_s_memrealtime();)
		constexpr int RU = 4;
		uint64_t nonempty = ballot64(cnt_l != 0);
		while (nonempty) {
			const int sl = __builtin_ctzll(nonempty);
			nonempty &= nonempty - 1;
			const uint32_t scnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, sl);
			const uint64_t ssb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(sb_l >> 32), sl) << 32) |
			    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sb_l, sl);
			RSTAT(n_cand += scnt;)
			for (uint32_t c0 = 0; c0 < scnt; c0 += WAVE * RU) {
				float scv[RU];
				uint32_t dcv[RU];
#pragma unroll
				for (int u = 0; u < RU; u++) {
					const uint32_t j = c0 + u * WAVE + lane;
					scv[u] = 0.0f;
					dcv[u] = 0;
					if (j < scnt) {
						const uint64_t at = ssb + j;
						scv[u] = A.cand_sc[at];
						dcv[u] = A.cand_doc ? A.cand_doc[at] : (uint32_t)at;
					}
				}
#pragma unroll
				for (int u = 0; u < RU; u++) {
					const bool valid = c0 + u * WAVE + lane < scnt;
					const float sc = scv[u];
					/* heap.c:68-74: when full, an item <= the root is dropped
					 * without touching the heap */
					uint64_t pend = ballot64(valid && (n < cap || sc > rmin));
					while (pend) {
						const int L = __builtin_ctzll(pend);
						const float v = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sc), L));
						const uint32_t dv = (uint32_t)__builtin_amdgcn_readlane((int)dcv[u], L);
						chp_add(h, K, n, cap, v, dv, rmin);
						RSTAT(n_ins++;)
						if (A.log_cnt && lane == 0) {
							const uint32_t nl = A.log_cnt[log_row];
							if (nl < A.log_cap) {
								A.log_ids[(uint64_t)log_row * A.log_cap + nl] = A.doc_ids[dv];
								A.log_sc[(uint64_t)log_row * A.log_cap + nl] = v;
							}
							A.log_cnt[log_row] = nl + 1;
						}
						pend &= pend - 1;
						pend &= ballot64(valid && (n < cap || sc > rmin));
					}
				}
			}
		}
	}
	RSTAT(const unsigned long long rt_heap = __builtin_amdgcn_s_memrealtime();)

	/* heap_sort (heap.c:197-221): repeated remove-min, placed from the back */
	const uint32_t cnt = n;
	while (n) {
		const uint32_t last = n - 1;
		const uint2 m = h[0];
		const uint32_t mx = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.x);
		const uint32_t my = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.y);
		chp_remove_min(h, K, n, rmin);
		h[last] = make_uint2(mx, my);
		asm volatile("" ::: "memory");
	}
#ifdef NXS_STATS
	if (lane == 0) {
		const unsigned long long rt4 = __builtin_amdgcn_s_memrealtime();
		atomicAdd(&g_rstats[0], 1ull);
		atomicAdd(&g_rstats[1], rt1 - rt0);
		atomicAdd(&g_rstats[2], rt_heap - rt1);
		atomicAdd(&g_rstats[3], rt4 - rt_heap);
		atomicMax(&g_rstats[4], rt4 - rt0);
		atomicAdd(&g_rstats[5], n_cand);
		atomicAdd(&g_rstats[6], n_ins);
		atomicMax(&g_rstats[7], n_ins * 1000000ull + n_cand / 16);
	}
#endif
	/* result i: node i of the sorted array */
	uint64_t *o_ids;
	float *o_sc;
	if (A.rec_base) {
		/* u32 count | u32 flags | u64 ids[k] | f32 scores[k]  (nxs_gpu.h) */
		uint8_t *rec = A.rec_base + (size_t)A.rec_slot[q] * A.rec_bytes;
		o_ids = (uint64_t *)(rec + 8);
		o_sc = (float *)(rec + 8 + 8 * (size_t)A.k);
		if (lane == 0) {
			((uint32_t *)rec)[0] = cnt;
		}
	} else {
		const uint64_t ob = A.out_off ? A.out_off[q] : (uint64_t)q * A.k;
		o_ids = A.out_ids + ob;
		o_sc = A.out_sc + ob;
		if (lane == 0) {
			A.out_count[q] = cnt;
		}
	}
	for (uint32_t i = lane; i < cnt; i += WAVE) {
		const uint2 e = h[i];
		o_ids[i] = A.doc_ids[e.y];
		o_sc[i] = __uint_as_float(e.x);
	}
}

/* ---- launcher ------------------------------------------------------- */

void
nxs_launch_replay(int heap, unsigned grid, size_t dyn_lds, hipStream_t st, const replay_args_t &r)
{
	switch (heap) {
	case HEAP_REG: hipLaunchKernelGGL(k_replay<HEAP_REG>, dim3(grid), dim3(WAVE), 0, st, r); break;
	case HEAP_LDS:
		if (r.flags & 1) {	/* NXS_GPU_OLDREPLAY: the one-lane form, for A/B runs */
			hipLaunchKernelGGL(k_replay<HEAP_LDS>, dim3(grid), dim3(WAVE), dyn_lds, st, r);
		} else {
			hipLaunchKernelGGL(k_replay_coop, dim3(grid), dim3(WAVE), dyn_lds, st, r);
		}
		break;
	default: hipLaunchKernelGGL(k_replay<HEAP_GLOBAL>, dim3(grid), dim3(WAVE), 0, st, r); break;
	}
}

/*
 * The merge step of the doc-sharded mode: the shards' accepted-candidate logs
 * of every query, highest shard (highest doc ids) first, are fed to the
 * reference's heap again (k_replay) -- by induction its state is the global
 * one (results.c:182-220 feeds descending doc id; heap.c:58-221).  Layout:
 * ids/scores [nq][n_shards][cap], counts [nq][n_shards], shard 0 = LOWEST docs.
 * Output: [nq][limit] + counts.  Runs on `device`, blocking.
 */
extern "C" int
nxsgpu_merge_candidates(int device, uint32_t limit, uint32_t nq, uint32_t n_shards, uint32_t cap,
    const uint64_t *ids, const float *scores, const uint32_t *counts,
    uint64_t *out_ids, float *out_scores, uint32_t *out_counts)
{
	const size_t nseg = (size_t)nq * n_shards, ncand = nseg * cap;
	std::vector<qmeta_t> qm(nq);
	void *ws = NULL;
	hipStream_t st = NULL;
	int rc = -1;

	if (limit == 0 || limit > NXSGPU_BIG_K) {
		set_error("nxsgpu_merge_candidates: limit must be 1..%d", NXSGPU_BIG_K);
		return -1;
	}
	if (nq == 0) {
		return 0;
	}
	for (size_t i = 0; i < nseg; i++) {
		if (counts[i] > cap) {
			set_error("nxsgpu_merge_candidates: a candidate log overflowed (%u > %u)", counts[i], cap);
			return -1;
		}
	}
	for (uint32_t q = 0; q < nq; q++) {
		qm[q].seg_first = q * n_shards;
		qm[q].n_groups = n_shards;
		qm[q].group_docs = 0;
		qm[q].pad = 0;
	}
	do {
		if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
			set_error("device setup failed");
			break;
		}
		const size_t need = 8192 + nq * sizeof(qmeta_t) + nseg * 4 + ncand * 12 + (size_t)nq * limit * 12 + nq * 4;
		if (hipMalloc(&ws, need) != hipSuccess) {
			set_error("hipMalloc(%zu) failed", need);
			break;
		}
		uint8_t *p = (uint8_t *)ws;
		qmeta_t *d_qm = carve<qmeta_t>(p, nq);
		uint32_t *d_cnt = carve<uint32_t>(p, nseg);
		uint64_t *d_ids = carve<uint64_t>(p, ncand);
		float *d_sc = carve<float>(p, ncand);
		uint64_t *d_oid = carve<uint64_t>(p, (size_t)nq * limit);
		float *d_osc = carve<float>(p, (size_t)nq * limit);
		uint32_t *d_ocnt = carve<uint32_t>(p, nq);
		replay_args_t ra;

		if (hipMemcpyAsync(d_qm, qm.data(), nq * sizeof(qmeta_t), hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemcpyAsync(d_cnt, counts, nseg * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemcpyAsync(d_ids, ids, ncand * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemcpyAsync(d_sc, scores, ncand * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
		    hipMemsetAsync(d_ocnt, 0, nq * 4, st) != hipSuccess) {
			set_error("upload failed");
			break;
		}
		memset(&ra, 0, sizeof(ra));
		ra.qmeta = d_qm;
		ra.seg_cap = cap;
		ra.seg_count = d_cnt;
		ra.cand_doc = NULL;		/* the candidate's index is its handle */
		ra.cand_sc = d_sc;
		ra.doc_ids = d_ids;
		ra.k = limit;
		ra.out_ids = d_oid;
		ra.out_sc = d_osc;
		ra.out_count = d_ocnt;
		/* the heap across the lanes (limit <= 64) or in LDS, worked on by the whole wavefront (k_replay_coop) */
		nxs_launch_replay(limit <= NXSGPU_FAST_K ? HEAP_REG : HEAP_LDS, nq, limit <= NXSGPU_FAST_K ? 0 : (size_t)limit * 8, st, ra);
		if (hipGetLastError() != hipSuccess ||
		    hipMemcpyAsync(out_ids, d_oid, (size_t)nq * limit * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipMemcpyAsync(out_scores, d_osc, (size_t)nq * limit * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipMemcpyAsync(out_counts, d_ocnt, nq * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
		    hipStreamSynchronize(st) != hipSuccess) {
			set_error("merge failed: %s", hipGetErrorString(hipGetLastError()));
			break;
		}
		rc = 0;
	} while (0);
	(void)hipFree(ws);
	if (st) {
		(void)hipStreamDestroy(st);
	}
	return rc;
}
