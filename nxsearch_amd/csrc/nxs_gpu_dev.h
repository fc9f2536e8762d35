/*
 * nxs_gpu_dev.h -- device-side helpers shared by the scan / replay kernels:
 * threshold hand-down between the wavefronts of a query, wave-level predicates
 * as scalar masks, the AGPR-held prefetch windows, 64-ary searches.
 */
#ifndef NXS_GPU_DEV_H
#define NXS_GPU_DEV_H

#include "nxs_gpu_int.h"

/*
 * Threshold hand-down between the wavefronts of one query.  A wavefront that
 * has finished its doc range publishes the k-th largest score it met (if it
 * met k).  The reference's heap is fed in descending doc id, so while range g
 * is being fed the heap root is at least the k-th largest score of ANY
 * higher range alone: a published value of a higher range is a valid
 * candidate threshold for range g from its very first doc.  Values only ever
 * make the filter tighter and a stale read (0 = nothing published yet, or an
 * older value in a non-coherent L2) is merely less tight -- correctness never
 * depends on visibility, so plain agent-scope relaxed accesses are enough.
 */
__device__ static inline float
range_hint(const scan_args_t &A, const qmeta_t &qm, uint32_t g)
{
	const unsigned lane = threadIdx.x & 63;
	float h = 0.0f;

	for (uint32_t g2 = g + 1 + lane; g2 < qm.n_groups; g2 += WAVE) {
		h = fmaxf(h, __hip_atomic_load(&A.pub[(uint64_t)qm.seg_first + g2],
		    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	}
	for (int o = 32; o; o >>= 1) {
		h = fmaxf(h, __shfl_xor(h, o));
	}
	return h;
}

__device__ static inline void
range_publish(const scan_args_t &A, uint64_t seg, float kth)
{
	if ((threadIdx.x & 63) == 0 && kth > 0.0f) {
		__hip_atomic_store(&A.pub[seg], kth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

/* lower bound of `doc` in post[lo, hi) by doc ordinal */
__device__ static inline uint64_t
post_lower_bound(const posting_t *__restrict__ post, uint64_t lo, uint64_t hi, uint64_t doc)
{
	while (lo < hi) {
		const uint64_t mid = lo + ((hi - lo) >> 1);
		if (post[mid].doc < doc) lo = mid + 1; else hi = mid;
	}
	return lo;
}

/* compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>) */
template <int I, int N, typename F>
__device__ __forceinline__ void
static_for_impl(F &&f)
{
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for_impl<I + 1, N>(f);
	}
}
/*
 * Wave-level predicates without the int round trip of HIP's __ballot(): a lane
 * condition becomes a 64-bit scalar mask (the v_cmp result itself), a scalar
 * mask becomes a lane condition again (it is used as the select mask), and a
 * lane's rank inside a mask is the two v_mbcnt instructions.
 */
static __device__ __forceinline__ uint64_t
ballot64(bool p)
{
	return __builtin_amdgcn_ballot_w64(p);
}

static __device__ __forceinline__ bool
lane_of(uint64_t wave_uniform_mask)
{
	return __builtin_amdgcn_inverse_ballot_w64(wave_uniform_mask);
}

static __device__ __forceinline__ uint32_t
lanes_below(uint64_t m)
{
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

template <int N, typename F>
__device__ __forceinline__ void
static_for(F &&f)
{
	static_for_impl<0, N>(f);
}

/*
 * The prefetched posting window of term slot T ("set B" of the one-window scan
 * paths) lives in two accumulation registers that only these asm blocks name.
 *
 * Why not ordinary variables: a load the compiler tracks is waited for as
 * soon as its value is copied, and taking B over into A at a rotation is such
 * a copy -- the wavefront stalled for a memory latency every 64 postings.  A
 * load it does not track (inline asm into a C variable) is not safe either:
 * the register allocator is free to move that variable with v_mov while the
 * data is still in flight.  The kernels use no AGPRs otherwise, so these
 * registers are out of the compiler's reach: nothing can be scheduled into, copied out
 * of, or reallocated over a pending prefetch (bpair_request / bpair_take below).
 */

/*
 * Prefetch RING of the tile path: R windows per term in flight instead of one.
 * A dense term drains a 64-posting window in one visit (a few hundred cycles)
 * while a window takes a memory latency (1-2 us under load) to arrive, so with
 * one window in flight every rotation of a dense term exposed that latency.
 * Term slot T owns the AGPR pairs [T*R, T*R + R); pair p holds the window that
 * is p-th to be consumed (mod R, `ring position`).  Taking the oldest window:
 *
 *     s_waitcnt vmcnt(R - 1)
 *
 * is exact enough and needs no bookkeeping: vector memory operations retire in
 * issue order, the R - 1 other pairs of this term were requested after the
 * oldest one (every take re-requests the pair it has just read, load_ring()
 * requests all R in order, addresses are clamped instead of predicated so the
 * count never varies), hence at most R - 1 operations outstanding means the
 * oldest has landed.  Loads of other terms issued in between only make the
 * wait longer than necessary, never shorter.  The position is wave-uniform
 * (an SGPR): the switch below is a scalar branch tree, once per 64 postings.
 */
template <int I> __device__ __forceinline__ void bpair_request(const posting_t *np);
template <int PAIR, int N> struct bpair_take_impl;
#define	NXS_BPAIR(I, RD, RI, RP)							\
template <> __device__ __forceinline__ void						\
bpair_request<I>(const posting_t *np)							\
{											\
	asm volatile(									\
	    "global_load_dwordx2 " RP ", %0, off"					\
	    : : "v"(np) : "memory", RD, RI);						\
}											\
template <int N> struct bpair_take_impl<I, N> {					\
	static __device__ __forceinline__ void						\
	run(uint32_t &ad, float &ai, const posting_t *np)				\
	{										\
		asm volatile(								\
		    "s_waitcnt vmcnt(%3)\n\t"						\
		    "v_accvgpr_read_b32 %0, " RD "\n\t"					\
		    "v_accvgpr_read_b32 %1, " RI "\n\t"					\
		    "global_load_dwordx2 " RP ", %2, off"				\
		    : "=&v"(ad), "=&v"(ai) : "v"(np), "i"(N) : "memory", RD, RI);	\
	}										\
};
NXS_BPAIR(0, "a0", "a1", "a[0:1]")
NXS_BPAIR(1, "a2", "a3", "a[2:3]")
NXS_BPAIR(2, "a4", "a5", "a[4:5]")
NXS_BPAIR(3, "a6", "a7", "a[6:7]")
NXS_BPAIR(4, "a8", "a9", "a[8:9]")
NXS_BPAIR(5, "a10", "a11", "a[10:11]")
NXS_BPAIR(6, "a12", "a13", "a[12:13]")
NXS_BPAIR(7, "a14", "a15", "a[14:15]")
NXS_BPAIR(8, "a16", "a17", "a[16:17]")
NXS_BPAIR(9, "a18", "a19", "a[18:19]")
NXS_BPAIR(10, "a20", "a21", "a[20:21]")
NXS_BPAIR(11, "a22", "a23", "a[22:23]")
NXS_BPAIR(12, "a24", "a25", "a[24:25]")
NXS_BPAIR(13, "a26", "a27", "a[26:27]")
NXS_BPAIR(14, "a28", "a29", "a[28:29]")
NXS_BPAIR(15, "a30", "a31", "a[30:31]")
NXS_BPAIR(16, "a32", "a33", "a[32:33]")
NXS_BPAIR(17, "a34", "a35", "a[34:35]")
NXS_BPAIR(18, "a36", "a37", "a[36:37]")
NXS_BPAIR(19, "a38", "a39", "a[38:39]")
NXS_BPAIR(20, "a40", "a41", "a[40:41]")
NXS_BPAIR(21, "a42", "a43", "a[42:43]")
NXS_BPAIR(22, "a44", "a45", "a[44:45]")
NXS_BPAIR(23, "a46", "a47", "a[46:47]")
NXS_BPAIR(24, "a48", "a49", "a[48:49]")
NXS_BPAIR(25, "a50", "a51", "a[50:51]")
NXS_BPAIR(26, "a52", "a53", "a[52:53]")
NXS_BPAIR(27, "a54", "a55", "a[54:55]")
NXS_BPAIR(28, "a56", "a57", "a[56:57]")
NXS_BPAIR(29, "a58", "a59", "a[58:59]")
NXS_BPAIR(30, "a60", "a61", "a[60:61]")
NXS_BPAIR(31, "a62", "a63", "a[62:63]")
#undef NXS_BPAIR
template <int I, int N> __device__ __forceinline__ void
bpair_take(uint32_t &ad, float &ai, const posting_t *np)
{
	bpair_take_impl<I, N>::run(ad, ai, np);
}

/*
 * Exact wait for one ring load.  Vector memory operations retire in issue
 * order, so a load is done once at most `younger` operations are outstanding,
 * `younger` = operations issued after it.  The kernels stamp every ring load
 * with a per-wavefront issue counter, which gives a lower bound of that number
 * (operations the compiler issues are not counted: the wait can only be longer
 * than needed, never shorter).  Waiting for vmcnt(R - 1) instead made a term
 * whose window rotates right after another term's wait for that term's brand
 * new request: a full memory latency.  s_waitcnt takes an immediate, hence the
 * branch tree; `younger` is wave-uniform.
 */
#define	VM_WAIT(n)	asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void
vm_wait_younger(uint32_t younger)
{
	if (younger >= 8) {
		if (younger >= 16) {
			VM_WAIT(16);
		} else if (younger >= 12) {
			VM_WAIT(12);
		} else if (younger >= 10) {
			VM_WAIT(10);
		} else {
			VM_WAIT(8);
		}
	} else if (younger >= 4) {
		if (younger >= 6) {
			if (younger == 7) { VM_WAIT(7); } else { VM_WAIT(6); }
		} else {
			if (younger == 5) { VM_WAIT(5); } else { VM_WAIT(4); }
		}
	} else if (younger >= 2) {
		if (younger == 3) { VM_WAIT(3); } else { VM_WAIT(2); }
	} else {
		if (younger == 1) { VM_WAIT(1); } else { VM_WAIT(0); }
	}
}

/* request pair `pos` of term slot T (R pairs per term) */
template <int T, int R> __device__ __forceinline__ void
bring_request(uint32_t pos, const posting_t *np)
{
	static_assert(R == 1 || R == 2 || R == 4 || R == 8, "ring depth");
	static_assert(T * R + R <= 32, "AGPR pairs");
	if constexpr (R == 1) {
		bpair_request<T>(np);
	} else {
		static_for<R>([&](auto rc) {
			constexpr int r = decltype(rc)::value;
			if (pos == (uint32_t)r) {
				bpair_request<T * R + r>(np);
			}
		});
	}
}

/* wait for pair `pos` (the oldest of term slot T), read it, re-request it */
template <int T, int R> __device__ __forceinline__ void
bring_take(uint32_t pos, uint32_t younger, uint32_t &ad, float &ai, const posting_t *np)
{
#ifndef NXS_VMWAIT_STAMPS
	(void)younger;
	vm_wait_younger(R - 1);		/* the stamp-free wait: the R - 1 siblings are younger */
#else
	vm_wait_younger(younger);
#endif
	if constexpr (R == 1) {
		bpair_take<T, 63>(ad, ai, np);
	} else {
		static_for<R>([&](auto rc) {
			constexpr int r = decltype(rc)::value;
			if (pos == (uint32_t)r) {
				bpair_take<T * R + r, 63>(ad, ai, np);
			}
		});
	}
}

/* issue stamps of a term's R ring loads, oldest first (FIFO) */
template <int R> struct ring_stamps {
	uint32_t st[R];
	/* operations issued after the oldest load of this ring */
	__device__ __forceinline__ uint32_t younger(uint32_t seq) const { return seq - st[0] - 1; }
	/* the oldest was consumed and requested again with stamp `seq` */
	__device__ __forceinline__ void rotate(uint32_t seq)
	{
#pragma unroll
		for (int r = 0; r + 1 < R; r++) {
			st[r] = st[r + 1];
		}
		st[R - 1] = seq;
	}
};

/*
 * Wave-cooperative lower bound: first index in [lo, hi) (relative to pt) whose
 * doc is >= bound, or hi.  64-ary: each round the 64 lanes probe 64 evenly
 * spaced postings, so a 10M-entry list needs 4 dependent loads, not 24.
 * All arguments and the result are wave-uniform.
 */
__device__ static inline int32_t
wave_lower_bound(const posting_t *__restrict__ pt, int32_t lo, int32_t hi, uint32_t bound)
{
	const int32_t lane = (int32_t)(threadIdx.x & 63);

	while (hi - lo > WAVE) {
		const int32_t step = (hi - lo + WAVE - 1) / WAVE;
		const int32_t idx = lo + lane * step;
		const bool valid = idx < hi;
		uint32_t v = 0xffffffffu;
		if (valid) {
			v = pt[idx].doc;
		}
		/* lanes are monotone: the first lane whose probe is >= bound */
		const uint64_t m = ballot64(!valid || v >= bound);
		const int32_t L = m ? (int32_t)__ffsll((long long)m) - 1 : WAVE;
		if (L == 0) {
			return lo;
		}
		const int32_t nlo = lo + (L - 1) * step + 1;
		const int32_t nhi = (L < WAVE && lo + L * step < hi) ? lo + L * step : hi;
		lo = nlo;
		hi = nhi;
	}
	{
		const int32_t idx = lo + lane;
		const bool valid = idx < hi;
		uint32_t v = 0;
		if (valid) {
			v = pt[idx].doc;
		}
		const uint64_t m = ballot64(valid && v >= bound);
		return m ? lo + (int32_t)__ffsll((long long)m) - 1 : hi;
	}
}

/* byte index of doc d's mask inside a tile: a u32 read at word (s*64+lane)
 * yields the four docs s*256 + j*64 + lane, j = 0..3 */
__device__ static inline uint32_t
mask_byte(uint32_t d)
{
	return ((d >> 8) << 8) | ((d & 63) << 2) | ((d >> 6) & 3);
}

/* evaluate the postfix boolean program on a presence mask */
__device__ static inline bool
eval_prog(const uint8_t *prog, uint32_t len, uint32_t m)
{
	uint64_t st = 0;	/* bit stack, top at bit 0 */
	for (uint32_t i = 0; i < len; i++) {
		const uint8_t op = prog[i];
		if (op < NXSGPU_MAX_TOKENS) {
			st = (st << 1) | ((m >> op) & 1);
		} else if (op == NXSGPU_OP_EMPTY) {
			st <<= 1;
		} else {
			const uint64_t b = st & 1, a = (st >> 1) & 1;
			uint64_t r;
			if (op == NXSGPU_OP_AND) r = a & b;
			else if (op == NXSGPU_OP_OR) r = a | b;
			else r = a & ~b & 1;
			st = ((st >> 2) << 1) | r;
		}
	}
	return st & 1;
}

/*
 * A workgroup here is ONE wavefront: its DS operations execute in issue order,
 * so cross-lane LDS hand-offs need no s_barrier -- and must not get one:
 * __syncthreads() also drains vmcnt(0), i.e. every posting prefetch in flight.
 * This only stops the compiler from moving memory operations across the point:
 * the wavefront-scope fence is what does that (it generates no instruction on
 * gfx9 -- one wavefront's LDS operations are ordered anyway -- but the optimiser
 * must treat it as a clobber of all memory: no store-to-load forwarding of a
 * lane's own LDS store across it, no load hoisted above another lane's store);
 * the wave barrier alone is declared IntrNoMem and promises nothing of the kind.
 */
#define	WAVE_SYNC()	do {							\
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");			\
	__builtin_amdgcn_wave_barrier();					\
} while (0)

#define	LIST_CAP	512
#ifndef SCAN8_RING_MAX
#define	SCAN8_RING_MAX	2		/* prefetch ring depth of the one-window tile path */
#endif
#ifndef SCAN8_RING_BIG
#define	SCAN8_RING_BIG	1		/* ... of its MODE_BIG instantiations: 92 + 20 registers are four wavefronts per SIMD, 82 + 10
					 * five -- a default-limit C3 batch, three alternating runs: 272.7 / 272.8 / 272.7 k queries/s
					 * at depth 2, 277.0 / 278.0 / 278.6 at 1, 252 at 4 */
#endif
#ifndef SCANR_RING
#define	SCANR_RING	4		/* prefetch ring depth of the required-term path (span rounds: 1, 2, 4 measured equal; whole-window rounds rotate the driver every round: 1 -> 4 = 1.22 -> 1.19 ms per C3 step) */
#endif
#ifndef SCANM_RING
#define	SCANM_RING	2		/* prefetch ring depth of the mask path */
#endif
#define	TCAND_CAP	64

/*
 * MODE_BIG: candidate threshold for 64 < k <= NXSGPU_BIG_K.
 *
 * The reference's heap drops an item iff it is full and the item is <= its root
 * (heap.c:68-74); the root is the k-th largest score fed so far.  The filter
 * needs a value that is NEVER ABOVE that root: any lower bound of the k-th
 * largest score among the docs this wavefront has emitted will do (they were
 * all fed before the doc being tested).  A histogram gives one without keeping
 * k scores: bucket(s) = the top bits of the float (sign 0, exponent, 4 mantissa
 * bits: 16 buckets per octave, 2^-7 .. 2^9, clamped) is monotone in s, so if the
 * buckets >= j together hold >= k emitted docs, the k-th largest emitted score
 * is >= the lower edge of bucket j -- exactly representable, no rounding
 * argument needed.  Docs that were not emitted (score <= an earlier threshold)
 * are not counted: the bound only gets weaker, never wrong.  Candidates = what
 * beats the edge; the replay applies the exact test.
 */
#ifndef BIGK_OCT_BITS
#define	BIGK_OCT_BITS	4		/* log2(buckets per octave): 16 (32: 2 KB of LDS, -2 % on a default-limit batch) */
#endif
#define	BIGK_BUCKETS	(16 << BIGK_OCT_BITS)	/* 16 octaves: 2^-7 .. 2^9 */
#define	BIGK_SH		(23 - BIGK_OCT_BITS)
#define	BIGK_BASE	((127 - 7) << BIGK_OCT_BITS)
#define	BIGK_PER_LANE	(BIGK_BUCKETS / WAVE)

static __device__ __forceinline__ uint32_t
bigk_bucket(float s)
{
	const int b = (int)(__float_as_uint(s) >> BIGK_SH) - BIGK_BASE;
	return (uint32_t)min(max(b, 0), BIGK_BUCKETS - 1);
}

/* lower edge of bucket j (bucket 0 reaches down to 0: "no threshold") */
static __device__ __forceinline__ float
bigk_edge(uint32_t j)
{
	return j ? __uint_as_float((j + BIGK_BASE) << BIGK_SH) : 0.0f;
}

/* count the emitted candidates of this step (lanes with `cand`) */
static __device__ __forceinline__ void
bigk_note(uint32_t *hist, bool cand, float sc)
{
	if (cand) {
		(void)__hip_atomic_fetch_add(&hist[bigk_bucket(sc)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	}
}

/*
 * Largest bucket edge with >= k counted docs at or above it (0 if there is none).
 * Lane L owns BIGK_PER_LANE consecutive buckets: its sum, a suffix sum over the lanes, the
 * highest lane whose suffix reaches k, then that lane's own eight counters.
 * Wave-uniform result.
 */
static __device__ __forceinline__ float
bigk_threshold(const uint32_t *hist, uint32_t k)
{
	const unsigned lane = threadIdx.x & 63;
	/* one wavefront's DS operations execute in issue order: the atomics above
	 * are visible; this only keeps the compiler from moving the reads up */
	asm volatile("" ::: "memory");
	uint32_t c[BIGK_PER_LANE];
	uint32_t own = 0;
#pragma unroll
	for (int i = 0; i < BIGK_PER_LANE; i += 4) {
		const uint4 q4 = *(const uint4 *)&hist[lane * BIGK_PER_LANE + i];
		c[i] = q4.x; c[i + 1] = q4.y; c[i + 2] = q4.z; c[i + 3] = q4.w;
	}
#pragma unroll
	for (int i = 0; i < BIGK_PER_LANE; i++) {
		own += c[i];
	}
	uint32_t suf = own;
#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) {
		const uint32_t v = (uint32_t)__shfl_down((int)suf, o);
		if (lane + o < WAVE) {
			suf += v;
		}
	}
	const uint64_t m = __builtin_amdgcn_ballot_w64(suf >= k);
	if (m == 0) {
		return 0.0f;
	}
	const int L = 63 - __builtin_clzll(m);		/* suffix sums do not increase with the lane */
	uint32_t acc = suf - own, j = 0;
	bool found = false;
#pragma unroll
	for (int i = BIGK_PER_LANE - 1; i >= 0; i--) {
		acc += c[i];
		if (!found && acc >= k) {
			j = lane * BIGK_PER_LANE + i;
			found = true;
		}
	}
	return bigk_edge((uint32_t)__builtin_amdgcn_readlane((int)j, L));
}

/*
 * The bookkeeping after a step has emitted `ne` candidates (wave-uniform; the
 * lanes with `cand` carry them): count them, and every `upd` candidates read the
 * threshold off the histogram again.
 */
static __device__ __forceinline__ void
bigk_account(uint32_t *hist, uint32_t k, uint32_t upd, bool cand, float sc, uint32_t ne,
    uint32_t &since, float hint, float &thr)
{
	bigk_note(hist, cand, sc);
	since += ne;
	if (since >= upd) {
		thr = fmaxf(hint, bigk_threshold(hist, k));
		since = 0;
	}
}

static __device__ __forceinline__ uint32_t
bigk_update_every(uint32_t k)
{
	return max(k >> 4, 32u);
}

/* the 64 lanes' values, sorted descending across the lanes (bitonic network) */
static __device__ __forceinline__ float
wave_sort_desc(float v)
{
	const unsigned lane = threadIdx.x & 63;
#pragma unroll
	for (unsigned k2 = 2; k2 <= WAVE; k2 <<= 1) {
#pragma unroll
		for (unsigned j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
			const float o = __shfl_xor(v, (int)j2);
			const bool desc = (lane & k2) == 0, lower = (lane & j2) == 0;
			v = (lower == desc) ? fmaxf(v, o) : fminf(v, o);
		}
	}
	return v;
}

/*
 * Threshold hand-down for limits > 64.  One range's own k-th best is a weak hint
 * when k is large (the best 1000 of the ~5000 matches of a range: its top fifth),
 * but the heap root while range g is fed is at least the k-th best of ALL higher
 * docs together.  A finished range therefore publishes lower bounds c_j of its
 * ceil(k / 2^j)-th best score, j = 0..5 (read off its histogram); if 2^j
 * DIFFERENT higher ranges each hold ceil(k / 2^j) docs scoring >= v, the docs
 * above range g hold k such docs and the root is >= v.  Every lane takes the
 * largest c_j over its own subset of the higher ranges (disjoint subsets:
 * different ranges), the 64 lane values are sorted, and the 2^j-th largest is
 * such a v.  The hint is the best over j.  Stale or partial reads only weaken it.
 */
#define	BIGK_SK		6

__device__ static inline float
bigk_hint(const scan_args_t &A, const qmeta_t &qm, uint32_t g)
{
	const unsigned lane = threadIdx.x & 63;
	float m[BIGK_SK], h = 0.0f;

#pragma unroll
	for (int j = 0; j < BIGK_SK; j++) {
		m[j] = 0.0f;
	}
	for (uint32_t g2 = g + 1 + lane; g2 < qm.n_groups; g2 += WAVE) {
		const float *p = A.pub_sk + ((uint64_t)qm.seg_first + g2) * 8;
#pragma unroll
		for (int j = 0; j < BIGK_SK; j++) {
			m[j] = fmaxf(m[j], __hip_atomic_load(p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		}
	}
	if (g + 1 >= qm.n_groups) {
		return 0.0f;		/* the top range: nothing above it */
	}
#pragma unroll
	for (int j = 0; j < BIGK_SK; j++) {
		const float s = wave_sort_desc(m[j]);
		h = fmaxf(h, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), (1 << j) - 1)));
	}
	return h;
}

__device__ static inline void
bigk_publish(const scan_args_t &A, uint64_t seg, const uint32_t *hist, uint32_t k)
{
#pragma unroll
	for (int j = 0; j < BIGK_SK; j++) {
		const float c = bigk_threshold(hist, (k + (1u << j) - 1) >> j);
		if ((threadIdx.x & 63) == 0 && c > 0.0f) {
			__hip_atomic_store(A.pub_sk + seg * 8 + j, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

#endif /* NXS_GPU_DEV_H */
