/*
 * nxs_gpu_fuzzy.hip -- BK-tree search on the device: match-first search + frontier search; nxsgpu_fuzzy
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include "nxs_gpu_dev.h"
#include "nxs_lev.h"

/* ------------------------------------------------------------------ */

struct fz_item_t { uint32_t tok, node; };

struct fz_args_t {
	const nxsgpu_bknode_t *	bk;
	const uint8_t *		bk_bytes;
	const uint8_t *		tok_bytes;
	const uint32_t *	tok_off;
	const uint64_t *	peq;		/* [n_tok][256] */
	const fz_item_t *	cur;
	fz_item_t *		next;
	const uint32_t *	cur_count;
	uint32_t *		next_count;
	uint32_t		cap;
	uint32_t *		best;		/* [n_tok] min BFS index of a usable match */
	unsigned long long *	visited;	/* [n_tok] or NULL */
	uint16_t *		dp_rows;	/* scratch for tokens > 64 bytes */
	uint32_t		dp_stride;
	uint32_t *		overflow;
	uint32_t		prune;		/* drop (token, node) pairs that can no longer win */
	unsigned long long *	evals;		/* distance evaluations (profiling) or NULL */
};

/* distance between token `tok` and the node's term */
__device__ static inline int
fz_distance(const fz_args_t &A, uint32_t tok, const nxsgpu_bknode_t &nd, uint64_t slot)
{
	const uint32_t qoff = A.tok_off[tok], m = A.tok_off[tok + 1] - qoff;
	const uint32_t n = nd.str_len;

	if (m == 0) {
		return (int)n;
	}
	if (m <= NXS_MYERS_MAXPAT) {
		/* Myers bit-vector, pattern = query token */
		const uint64_t *peq = A.peq + (uint64_t)tok * 256;
		nxs_myers_t s;
		nxs_myers_init(&s, m);
		/*
		 * Eight term bytes at a time: their Peq words are eight independent
		 * gathers issued together (one L2 round trip), then the dependent
		 * bit-vector steps.  Fetched inside the step loop they cost one round
		 * trip per byte -- the kernel was bound by exactly that latency.  Bytes
		 * past the term's end index a valid table row and are not stepped.
		 */
		uint64_t w;
		memcpy(&w, nd.inl, 8);
		const uint8_t *rest = A.bk_bytes + nd.str_off;
		for (uint32_t i0 = 0; i0 < n; i0 += 8) {
			uint64_t e[8];
			if (i0) {
				/* (the byte pool carries 16 bytes of slack behind its end) */
				uint32_t lo32, hi32;
				__builtin_memcpy(&lo32, rest + i0, 4);
				__builtin_memcpy(&hi32, rest + i0 + 4, 4);
				w = (uint64_t)lo32 | ((uint64_t)hi32 << 32);
			}
#pragma unroll
			for (int i = 0; i < 8; i++) {
				e[i] = peq[(w >> (8 * i)) & 0xff];
			}
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (i0 + i < n) {
					nxs_myers_step(&s, e[i]);
				}
			}
		}
		return s.score;
	}
	/* long token: row DP (levdist.c:67-150) in global scratch */
	{
		const uint8_t *a = A.tok_bytes + qoff;		/* length m */
		const uint8_t *b = A.bk_bytes + nd.str_off;	/* length n */
		uint16_t *row = A.dp_rows + slot * A.dp_stride;
		uint32_t la = m, lb = n;
		if (la < lb) {
			const uint8_t *t = a; a = b; b = t;
			const uint32_t tl = la; la = lb; lb = tl;
		}
		if (lb == 0) {
			return (int)la;
		}
		for (uint32_t j = 0; j <= lb; j++) {
			row[j] = (uint16_t)j;
		}
		for (uint32_t i = 0; i < la; i++) {
			uint32_t diag = i, above;
			row[0] = (uint16_t)(i + 1);
			for (uint32_t j = 1; j <= lb; j++) {
				above = row[j];
				uint32_t v = diag + (a[i] != b[j - 1]);
				v = min(v, (uint32_t)row[j - 1] + 1);
				v = min(v, above + 1);
				row[j] = (uint16_t)v;
				diag = above;
			}
		}
		return (int)row[lb];
	}
}

template <bool LONG>
__global__ void
k_bk_level(const fz_args_t A)
{
	__shared__ uint32_t s_wtot[16], s_base;
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint32_t count = min(*A.cur_count, A.cap);
	const uint32_t nthreads = gridDim.x * blockDim.x;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t rounds = (count + nthreads - 1) / nthreads;
	uint32_t n_eval = 0;

	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t i = r * nthreads + tid;
		uint32_t nkids = 0, first = 0, tok = 0;
		uint64_t bm = 0, full = 0;

		if (i < count) {
			const fz_item_t it = A.cur[i];
			const uint32_t m = A.tok_off[it.tok + 1] - A.tok_off[it.tok];
			tok = it.tok;
			/*
			 * Exact pruning.  The answer is the match of LOWEST BFS rank
			 * (idxterm.c:238-242: first pushed with total > 0; Q7), a node's
			 * descendants all have higher ranks than the node itself (BFS
			 * numbering), and best[tok] only ever decreases.  So once a usable
			 * match of rank r is known, a pair whose node has rank > r can
			 * neither be nor lead to the winner: it is dropped without a distance
			 * computation and without children.  Any value read here -- stale or
			 * written by a concurrent lane of this very level -- is the rank of a
			 * real match, hence a valid bound.  (Off when the caller wants the
			 * reference's visit counts.)
			 */
			const bool dead = A.prune && __hip_atomic_load(&A.best[it.tok], __ATOMIC_RELAXED,
			    __HIP_MEMORY_SCOPE_AGENT) < it.node;
			if (!dead && LONG == (m > NXS_MYERS_MAXPAT)) {
				const nxsgpu_bknode_t nd = A.bk[it.node];
				const int d = fz_distance(A, it.tok, nd, tid);
				n_eval++;
				if (A.visited) {
					atomicAdd(&A.visited[it.tok], 1ull);
				}
				/* match: bktree.c:252-254; winner = first pushed with
				 * total > 0 (idxterm.c:238-242) = min BFS index */
				if (d <= 2 && (nd.flags & 1)) {
					atomicMin(&A.best[it.tok], it.node);
				}
				/* children in slots [max(d-2,0), min(d+2,63)):
				 * bktree.c:150-156,260-264 (x86 shift semantics) */
				const unsigned min_d = d > 2 ? (unsigned)d - 2 : 0;
				const unsigned max_d = min((unsigned)d + 2, 63u);
				const uint64_t lo_mask = ~0ull << (min_d & 63);
				const uint64_t hi_mask = ~0ull >> ((64 - max_d) & 63);
				full = nd.bitmap;
				bm = full & lo_mask & hi_mask;
				nkids = __popcll(bm);
				first = nd.first_child;
			}
		}
		/* wave-level inclusive scan of nkids, one atomic per wavefront */
		uint32_t incl = nkids;
		for (int o = 1; o < WAVE; o <<= 1) {
			const uint32_t v = __shfl_up((int)incl, o);
			if (lane >= (unsigned)o) incl += v;
		}
		const uint32_t total = __shfl((int)incl, WAVE - 1);
		/*
		 * One returning atomic per WORKGROUP and round, not per wavefront: a
		 * single counter word takes ~88 M atomics/s (MI355X_MICROARCH.md,
		 * `dequeue`), and with one per 64 candidates that ceiling -- not memory,
		 * not the DP -- was the 5.4 G candidates/s this kernel ran at.  `rounds`
		 * is uniform over the grid, so every wavefront reaches the barriers.
		 */
		s_wtot[wid] = total;		/* (all lanes write the same value) */
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t sum = 0;
			for (unsigned w = 0; w < nw; w++) {
				const uint32_t tw = s_wtot[w];
				s_wtot[w] = sum;
				sum += tw;
			}
			s_base = sum ? atomicAdd(A.next_count, sum) : 0;
		}
		__syncthreads();
		const uint32_t wbase = s_base + s_wtot[wid];
		__syncthreads();		/* s_wtot is rewritten next round */
		uint32_t o = wbase + incl - nkids;
		/* ascending slot order = the order bktree_search pushes children */
		while (bm) {
			const int slot = __ffsll((long long)bm) - 1;
			bm &= bm - 1;
			const uint32_t child = first + __popcll(full & ((1ull << slot) - 1));
			if (o < A.cap) {
				fz_item_t ni;
				ni.tok = tok;
				ni.node = child;
				A.next[o] = ni;
			} else {
				*A.overflow = 1;
			}
			o++;
		}
	}
	if (A.evals) {
		for (int o = 32; o; o >>= 1) {
			n_eval += (uint32_t)__shfl_xor((int)n_eval, o);
		}
		if (lane == 0 && n_eval) {
			atomicAdd(A.evals, (unsigned long long)n_eval);
		}
	}
}

__global__ void
k_bk_seed(fz_item_t *items, uint32_t *count0, uint32_t n_tok, uint32_t *best,
    unsigned long long *visited)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_tok) {
		fz_item_t it;
		it.tok = i;
		it.node = 0;
		items[i] = it;
		best[i] = 0xffffffffu;
		if (visited) {
			visited[i] = 0;
		}
	}
	if (i == 0) {
		*count0 = n_tok;
	}
}

__global__ void
k_bk_peq(const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tok, uint64_t *peq, uint2 *tokf,
    const uint32_t *tok_rank)
{
	const uint32_t tok = blockIdx.x;
	const uint32_t off = tok_off[tok], m = tok_off[tok + 1] - off;
	if (tokf && threadIdx.x == 0) {
		/* what k_fz_filter compares: the token's byte set, at the token's place in
		 * the length-sorted order */
		uint32_t sg = 0;
		for (uint32_t j = 0; j < m; j++) {
			sg |= 1u << (tok_bytes[off + j] & 31);
		}
		tokf[tok_rank[tok]] = make_uint2(sg, tok);
	}
	for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) {
		uint64_t bits = 0;
		if (m <= NXS_MYERS_MAXPAT) {
			for (uint32_t j = 0; j < m; j++) {
				if (tok_bytes[off + j] == c) {
					bits |= 1ull << j;
				}
			}
		}
		peq[(uint64_t)tok * 256 + c] = bits;
	}
}

__global__ void
k_bk_finish(const nxsgpu_bknode_t *bk, const uint32_t *best, uint32_t n_tok, uint32_t *term_ids)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_tok) {
		const uint32_t b = best[i];
		term_ids[i] = (b == 0xffffffffu) ? 0 : bk[b].term_id;
	}
}

/* ---- match-first fuzzy search ---------------------------------------- */
/*
 * The level-by-level search above spends its time on the frontier: the winner
 * sits 9-12 levels deep and everything above it has to be expanded (65 M pairs
 * for 1024 tokens over a 1M-term tree).  But the winner has a closed form:
 *
 *   the node of LOWEST BFS rank among those that (1) are a match -- distance
 *   <= 2, on-disk total > 0 -- and (2) bktree_search reaches: at EVERY ancestor
 *   a the slot of the path's child lies in [max(d(q,a)-2, 0), min(d(q,a)+2, 63))
 *   (bktree.c:150-156,260-264; Q8: the range is half-open, so a match is not
 *   always reached).
 *
 * (1) needs no tree: all (token, term) pairs are screened with a necessary
 * condition -- |len difference| <= 2 and, on the sets of bytes the strings
 * contain (hashed to 64 bits), at most 2 bytes on either side that the other
 * string lacks: an edit removes at most one such byte per side -- 0.03-0.6 % of
 * the pairs survive on the synthetic vocabulary and take the exact bit-vector
 * distance.  (2) walks the few real matches up to the root.  The three steps
 * are three launches over flat queues; their result is the same min-rank node
 * (tests and bench.py compare against the level-by-level search with and
 * without pruning, and against the oracle).
 */
#define	FZ_NOPARENT	0xffffffffu
#define	FZF_BUF		192		/* survivors a wavefront stages in LDS */
#define	FZ_MAXLEN	(NXS_MYERS_MAXPAT + 2)	/* longest term that can be within 2 of a token */
#define	FZ_NQ		64		/* survivor sub-queues: a single counter word takes ~88 M atomics/s */
#define	FZ_CSTRIDE	16		/* their counters, one per 64 bytes */

/* per node: its parent and the slot it hangs in */
__global__ void __launch_bounds__(256)
k_bk_aux(const nxsgpu_bknode_t *bk, uint32_t n, uint32_t *parent, uint8_t *slot)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) {
		return;
	}
	if (i == 0) {
		parent[0] = FZ_NOPARENT;
		slot[0] = 0;
	}
	uint64_t bm = bk[i].bitmap;
	uint32_t c = bk[i].first_child;
	while (bm) {
		const int sl = __ffsll((long long)bm) - 1;
		bm &= bm - 1;
		parent[c] = i;
		slot[c] = (uint8_t)sl;
		c++;
	}
}

/* per candidate (the nodes that can win, sorted by term length on the host): the
 * set of bytes its term contains, hashed to 32 bits, and the length */
__global__ void __launch_bounds__(256)
k_fz_sigs(const nxsgpu_bknode_t *bk, const uint8_t *bytes, const uint32_t *cand_node, uint32_t n_c,
    uint32_t *sig, uint8_t *len8)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_c) {
		return;
	}
	const nxsgpu_bknode_t nd = bk[cand_node[i]];
	const uint8_t *str = bytes + nd.str_off;
	uint32_t sg = 0;
	for (uint32_t j = 0; j < nd.str_len; j++) {
		sg |= 1u << (str[j] & 31);
	}
	sig[i] = sg;
	len8[i] = (uint8_t)nd.str_len;
}

/*
 * Screen.  lane = one candidate node (sorted by length: a workgroup's 256 terms
 * span lengths [Lmin, Lmax]), loop = the tokens of length Lmin-2 .. Lmax+2
 * (tokens sorted by length too; their features are wave-uniform and come
 * through the scalar unit, four tokens per round).  grid.y slices the token
 * range.  A pair that passes wrongly (|length difference| = 3 across a length
 * boundary of the workgroup, hash collisions) is dropped by the exact distance.
 */
__global__ void __launch_bounds__(256)
k_fz_filter(const uint32_t *__restrict__ sig, const uint32_t *__restrict__ cand_node,
    const uint8_t *__restrict__ len8, uint32_t n_c, const uint2 *__restrict__ tokf,
    const uint32_t *__restrict__ tok_len_off, fz_item_t *out, uint32_t *out_count, uint32_t qcap,
    uint32_t *overflow)
{
	__shared__ fz_item_t s_buf[4][FZF_BUF];
	/* this workgroup's sub-queue: [sq * qcap, (sq + 1) * qcap) */
	const uint32_t sq = (blockIdx.x + 5 * blockIdx.y) & (FZ_NQ - 1);
	fz_item_t *const sq_out = out + (uint64_t)sq * qcap;
	uint32_t *const sq_count = out_count + sq * FZ_CSTRIDE;
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const uint32_t first = blockIdx.x * 256, i = first + threadIdx.x;
	const bool valid = i < n_c;
	const uint32_t ts = valid ? sig[i] : 0, nts = ~ts;
	const uint32_t node = valid ? cand_node[i] : 0;
	const uint64_t vmask = ballot64(valid);
	const uint32_t lmin = len8[first], lmax = len8[min(first + 255, n_c - 1)];
	const uint32_t ta = tok_len_off[lmin > 2 ? lmin - 2 : 0];
	const uint32_t tb = tok_len_off[min(lmax + 2, (uint32_t)NXS_MYERS_MAXPAT) + 1];
	const uint32_t per = (tb - ta + gridDim.y - 1) / gridDim.y;
	const uint32_t t0 = ta + blockIdx.y * per, t1 = min(tb, t0 + per);
	uint32_t nb = 0;

	auto flush = [&]() {
		uint32_t base = 0;
		if (lane == 0) {
			base = atomicAdd(sq_count, nb);
		}
		base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
		WAVE_SYNC();
		for (uint32_t e = lane; e < nb; e += WAVE) {
			if (base + e < qcap) {
				sq_out[base + e] = s_buf[wid][e];
			} else {
				*overflow = 1;
			}
		}
		WAVE_SYNC();
		nb = 0;
	};
	auto check = [&](const uint2 f, bool in_range) {
		/* bytes only the term has / only the token has: an edit removes at
		 * most one of each */
		const uint32_t a = __popc(ts & ~f.x), b = __popc(nts & f.x);
		const uint64_t m = in_range ? (ballot64(max(a, b) <= 2u) & vmask) : 0ull;
		if (m) {
			if (lane_of(m)) {
				fz_item_t it;
				it.tok = f.y;
				it.node = node;
				s_buf[wid][nb + lanes_below(m)] = it;
			}
			nb += __popcll(m);
			if (nb > FZF_BUF - WAVE) {
				flush();
			}
		}
	};
	for (uint32_t j = t0; j < t1; j += 4) {
		/* (the array carries four entries of slack behind its end) */
		const uint2 f0 = tokf[j], f1 = tokf[j + 1], f2 = tokf[j + 2], f3 = tokf[j + 3];
		check(f0, true);
		check(f1, j + 1 < t1);
		check(f2, j + 2 < t1);
		check(f3, j + 3 < t1);
	}
	if (nb) {
		flush();
	}
}

/*
 * Exact distance of the screened pairs (grid.y = sub-queue).  A match at
 * distance <= 1 is always reached: every node of a child's subtree is at the
 * child's slot distance s from the ancestor a (slot-63 subtrees, which no search
 * ever enters, are not candidates), so |d(q,a) - s| <= 1 and s lies inside
 * [d(q,a)-2, d(q,a)+2) at every ancestor -- it lowers best[] right here.  A
 * match at distance 2 misses exactly when some ancestor has d(q,a) = s - 2: it
 * goes to the next queue for the walk (one returning atomic per workgroup and
 * round, as in k_bk_level).
 */
__global__ void __launch_bounds__(1024)
k_fz_dist(const fz_args_t A, const fz_item_t *cand, const uint32_t *cand_count, uint32_t qcap, fz_item_t *match,
    uint32_t *match_count, uint32_t mcap)
{
	__shared__ uint32_t s_wtot[16], s_base;
	const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint32_t count = min(cand_count[blockIdx.y * FZ_CSTRIDE], qcap);
	const uint32_t nthreads = gridDim.x * blockDim.x;
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t rounds = (count + nthreads - 1) / nthreads;
	uint32_t n_eval = 0;

	cand += (uint64_t)blockIdx.y * qcap;
	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t i = r * nthreads + tid;
		fz_item_t it;
		bool hit = false;

		it.tok = it.node = 0;
		if (i < count) {
			it = cand[i];
			const nxsgpu_bknode_t nd = A.bk[it.node];
			const int d = fz_distance(A, it.tok, nd, 0);
			n_eval++;
			if (d <= 1) {
				atomicMin(&A.best[it.tok], it.node);
			} else if (d == 2) {	/* bktree.c:252-254 */
				hit = __hip_atomic_load(&A.best[it.tok], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > it.node;
			}
		}
		const uint64_t m = ballot64(hit);
		s_wtot[wid] = __popcll(m);
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t sum = 0;
			for (unsigned w = 0; w < nw; w++) {
				const uint32_t tw = s_wtot[w];
				s_wtot[w] = sum;
				sum += tw;
			}
			s_base = sum ? atomicAdd(match_count, sum) : 0;
		}
		__syncthreads();
		const uint32_t o = s_base + s_wtot[wid] + lanes_below(m);
		__syncthreads();
		if (hit) {
			if (o < mcap) {
				match[o] = it;
			} else {
				*A.overflow = 1;
			}
		}
	}
	if (A.evals) {
		for (int o = 32; o; o >>= 1) {
			n_eval += (uint32_t)__shfl_xor((int)n_eval, o);
		}
		if (lane == 0 && n_eval) {
			atomicAdd(A.evals, (unsigned long long)n_eval);
		}
	}
}

/* does bktree_search reach the match?  Walk to the root; every ancestor's child
 * range must hold the slot the path leaves it through. */
__global__ void __launch_bounds__(256)
k_fz_chain(const fz_args_t A, const uint32_t *__restrict__ parent, const uint8_t *__restrict__ slot,
    const fz_item_t *match, const uint32_t *match_count, uint32_t mcap)
{
	const unsigned lane = threadIdx.x & 63;
	const uint32_t count = min(*match_count, mcap);
	const uint32_t nthreads = gridDim.x * blockDim.x;
	uint32_t n_eval = 0;

	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += nthreads) {
		const fz_item_t it = match[i];
		uint32_t c = it.node;
		bool ok = true;

		for (;;) {
			/* (a match of lower rank is known: this one cannot win -- any value
			 * read is the rank of a reachable match) */
			if (__hip_atomic_load(&A.best[it.tok], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it.node) {
				ok = false;
				break;
			}
			const uint32_t p = parent[c];
			if (p == FZ_NOPARENT) {
				break;
			}
			const unsigned sl = slot[c];
			const nxsgpu_bknode_t nd = A.bk[p];
			const int d = fz_distance(A, it.tok, nd, 0);
			n_eval++;
			/* bktree.c:150-156,260-264 (x86 shift semantics), as k_bk_level */
			const unsigned min_d = d > 2 ? (unsigned)d - 2 : 0;
			const unsigned max_d = min((unsigned)d + 2, 63u);
			const uint64_t lo_mask = ~0ull << (min_d & 63);
			const uint64_t hi_mask = ~0ull >> ((64 - max_d) & 63);
			if (!(((lo_mask & hi_mask) >> sl) & 1)) {
				ok = false;
				break;
			}
			c = p;
		}
		if (ok) {
			atomicMin(&A.best[it.tok], it.node);
		}
	}
	if (A.evals) {
		for (int o = 32; o; o >>= 1) {
			n_eval += (uint32_t)__shfl_xor((int)n_eval, o);
		}
		if (lane == 0 && n_eval) {
			atomicAdd(A.evals, (unsigned long long)n_eval);
		}
	}
}

/* ------------------------------------------------------------------ */
/* host side of the shim                                               */

void
bk_aux_free(nxsgpu_index_t *ix)
{
	(void)hipFree(ix->d_bk_parent);
	(void)hipFree(ix->d_bk_slot);
	(void)hipFree(ix->d_fz_node);
	(void)hipFree(ix->d_fz_sig);
	(void)hipFree(ix->d_fz_len);
	ix->d_bk_parent = NULL;
	ix->d_bk_slot = NULL;
	ix->d_fz_node = NULL;
	ix->d_fz_sig = NULL;
	ix->d_fz_len = NULL;
	ix->n_fz = 0;
}

/* the match-first search's view of the tree (after d_bk / d_bk_bytes are in place) */
int
bk_aux_build(nxsgpu_index_t *ix, const nxsgpu_bknode_t *nodes, uint32_t n)
{
	bk_aux_free(ix);
	if (n == 0) {
		return 0;
	}
	/* candidates: nodes with postings on disk whose term can be within 2 of a
	 * token of <= 64 bytes; counting sort by length, BFS rank inside a length */
	std::vector<uint32_t> start(FZ_MAXLEN + 2, 0), perm;
	/* (a child in slot 63 is never visited -- the range's upper end is at most 63,
	 * exclusive: bktree.c:150-156 -- and neither is anything below it; BFS
	 * numbering: a parent precedes its children) */
	std::vector<uint8_t> cut(n, 0);
	uint32_t n_c = 0;
	for (uint32_t i = 0; i < n; i++) {
		uint64_t bm = nodes[i].bitmap;
		uint32_t c = nodes[i].first_child;
		while (bm) {
			const int sl = __builtin_ctzll(bm);
			bm &= bm - 1;
			if (c < n) {
				cut[c] = cut[i] | (sl >= 63);
			}
			c++;
		}
	}
	auto is_cand = [&](uint32_t i) { return (nodes[i].flags & 1) && nodes[i].str_len <= FZ_MAXLEN && !cut[i]; };
	for (uint32_t i = 0; i < n; i++) {
		if (is_cand(i)) {
			start[nodes[i].str_len + 1]++;
			n_c++;
		}
	}
	for (uint32_t l = 0; l <= FZ_MAXLEN; l++) {
		start[l + 1] += start[l];
	}
	ix->fz_len_start.assign(start.begin(), start.end());	/* first candidate of every length */
	perm.resize(std::max<uint32_t>(n_c, 1));
	for (uint32_t i = 0; i < n; i++) {
		if (is_cand(i)) {
			perm[start[nodes[i].str_len]++] = i;
		}
	}
	if (hipMalloc(&ix->d_bk_parent, (size_t)n * 4) != hipSuccess ||
	    hipMalloc(&ix->d_bk_slot, (size_t)n + 16) != hipSuccess ||
	    hipMalloc(&ix->d_fz_node, (size_t)std::max<uint32_t>(n_c, 1) * 4) != hipSuccess ||
	    hipMalloc(&ix->d_fz_sig, (size_t)std::max<uint32_t>(n_c, 1) * 4) != hipSuccess ||
	    hipMalloc(&ix->d_fz_len, (size_t)n_c + 16) != hipSuccess ||
	    hipMemcpyAsync(ix->d_fz_node, perm.data(), (size_t)n_c * 4, hipMemcpyHostToDevice, ix->stream_fz) != hipSuccess) {
		bk_aux_free(ix);
		set_error("BK-tree side arrays: out of device memory");
		return -1;
	}
	hipLaunchKernelGGL(k_bk_aux, dim3((n + 255) / 256), dim3(256), 0, ix->stream_fz,
	    ix->d_bk, n, ix->d_bk_parent, ix->d_bk_slot);
	if (n_c) {
		hipLaunchKernelGGL(k_fz_sigs, dim3((n_c + 255) / 256), dim3(256), 0, ix->stream_fz,
		    ix->d_bk, ix->d_bk_bytes, ix->d_fz_node, n_c, ix->d_fz_sig, ix->d_fz_len);
	}
	if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream_fz) != hipSuccess) {
		bk_aux_free(ix);
		set_error("k_bk_aux failed");
		return -1;
	}
	ix->n_fz = n_c;
	return 0;
}

/* ---- fuzzy ----------------------------------------------------------- */

/*
 * Match-first search of all tokens at once (tokens of <= 64 bytes only), in two halves: mf_launch
 * queues the upload, the kernels and the copy back on stream_fz and returns (0 / -1); mf_finish waits
 * for them: 0 = term_ids filled, 1 = a queue overflowed (the caller takes the level-by-level
 * search), -1 = error.  Between the two the pass owns the fuzzy workspaces (fz, fz_pin).
 */
static int
mf_launch(nxsgpu_index_t *ix, int slot, const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tok)
{
	nxsgpu_index::fz_slot_t &sl = ix->fzs[slot];
	const uint32_t n_c = ix->n_fz;
	const uint32_t blen = tok_off[n_tok] - tok_off[0];
	/* FZ_NQ sub-queues of qcap survivors each */
	/* (what a sub-queue can receive at most: its workgroups x 256 nodes x the
	 * tokens of a grid.y slice -- a small tree fills few sub-queues) */
	const uint32_t gy = std::max<uint32_t>(1, std::min<uint32_t>(8, n_tok / 128));
	const uint64_t q_most = (uint64_t)gy * (((n_c + 255) / 256 + FZ_NQ - 1) / FZ_NQ) * 256 * ((n_tok + gy - 1) / gy);
	const uint64_t qcap = std::min<uint64_t>(0xffffffffu, std::max<uint64_t>((n_tok + FZ_NQ - 1) / FZ_NQ + 1,
	    std::min<uint64_t>(ix->cfg.fuzzy_cand / FZ_NQ, q_most)));
	const uint64_t ccap = qcap * FZ_NQ;
	const uint64_t mcap = std::max<uint64_t>(1024, ccap / 4);
	const size_t need = 16384 + FZ_NQ * FZ_CSTRIDE * 4 + (ccap + mcap) * sizeof(fz_item_t) + (size_t)n_tok * (256 * 8 + 8 + 4 + 4 + 4) + 64 + blen + 16 +
	    ((size_t)n_tok + 1) * 4 + (NXS_MYERS_MAXPAT + 4) * 4 + 16 * 256 + 256;
	/* ONE upload from pinned memory -- token offsets, rank of every token in the length-sorted order, first
	 * rank of every length, then the tokens' bytes -- and ONE copy back (term ids, counters, evaluation counts):
	 * a copy to or from pageable memory is staged by the runtime and costs the host 20-40 us apiece, and the
	 * caller (plan_batch) sits waiting for this pass */
	const size_t up_words = (size_t)n_tok + 1 + n_tok + NXS_MYERS_MAXPAT + 2;
	const size_t up_bytes = (up_words * 4 + 15) & ~(size_t)15;
	const size_t dn_bytes = ((size_t)n_tok * 4 + 15 & ~(size_t)15) + 16 + 16;
	const size_t qcnt_bytes = (size_t)FZ_NQ * FZ_CSTRIDE * 4;	/* (copied back with the rest when profiling) */
	const size_t pin_need = up_bytes + blen + 16 + dn_bytes + qcnt_bytes + 64;
	if (sl.pin_len < pin_need) {
		if (sl.pin) {
			(void)hipHostFree(sl.pin);
			sl.pin = NULL;
			sl.pin_len = 0;
		}
		if (hipHostMalloc((void **)&sl.pin, pin_need + pin_need / 2, hipHostMallocDefault) != hipSuccess) {
			set_error("hipHostMalloc(%zu) for the fuzzy staging failed", pin_need);
			return -1;
		}
		sl.pin_len = pin_need + pin_need / 2;
	}
	uint32_t *const up = (uint32_t *)sl.pin;
	uint8_t *const h_dn = sl.pin + up_bytes + ((blen + 16 + 15) & ~(size_t)15);
	uint32_t *roff = up, *rank = roff + n_tok + 1, *len_off = rank + n_tok;

	if (sl.ws_len < need) {
		if (sl.ws) {
			(void)hipFree(sl.ws);
			sl.ws = NULL;
			sl.ws_len = 0;
		}
		if (hipMalloc(&sl.ws, need) != hipSuccess) {
			set_error("hipMalloc(%zu) for the fuzzy workspace failed", need);
			return -1;
		}
		sl.ws_len = need;
	}
	uint8_t *p = (uint8_t *)sl.ws;
	fz_item_t *d_cand = carve<fz_item_t>(p, ccap);
	fz_item_t *d_match = carve<fz_item_t>(p, mcap);
	/* (what comes back, in one piece: term ids | counters | evaluation counts; the sub-queue counters behind
	 * them are zeroed with the same memset) */
	uint8_t *d_dn = carve<uint8_t>(p, dn_bytes + FZ_NQ * FZ_CSTRIDE * 4);
	uint32_t *d_tids = (uint32_t *)d_dn;
	uint32_t *d_cnt = (uint32_t *)(d_dn + (((size_t)n_tok * 4 + 15) & ~(size_t)15));	/* -, matches, overflow, (seed's count) */
	unsigned long long *d_evals = (unsigned long long *)(d_cnt + 4);	/* distance evaluations, pairs compared */
	uint32_t *d_qcnt = (uint32_t *)(d_evals + 2);		/* survivors per sub-queue */
	uint64_t *d_peq = carve<uint64_t>(p, (size_t)n_tok * 256);
	uint2 *d_tokf = carve<uint2>(p, (size_t)n_tok + 4);
	uint32_t *d_best = carve<uint32_t>(p, n_tok);
	/* (what goes up, in one piece: the offset / rank / length words, then the bytes) */
	uint8_t *d_upb = carve<uint8_t>(p, up_bytes + blen + 16);
	uint32_t *d_up = (uint32_t *)d_upb;
	uint8_t *d_bytes = d_upb + up_bytes;
	uint32_t *d_off = d_up, *d_rank = d_up + n_tok + 1, *d_len_off = d_rank + n_tok;
	hipStream_t st = ix->stream_fz;
	fz_args_t fa;

#ifdef NXS_DBG_FZT
	static double acc[8]; static int ncall;
	auto nowd = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
	double tt = nowd();
#define FZT(i) do { double n_ = nowd(); acc[i] += n_ - tt; tt = n_; } while (0)
#else
#define FZT(i) do { } while (0)
#endif
	for (uint32_t i = 0; i <= n_tok; i++) {
		roff[i] = tok_off[i] - tok_off[0];
	}
	for (uint32_t l = 0; l <= NXS_MYERS_MAXPAT + 1; l++) {
		len_off[l] = 0;
	}
	for (uint32_t i = 0; i < n_tok; i++) {
		len_off[roff[i + 1] - roff[i] + 1]++;		/* (every token is <= 64 bytes here) */
	}
	for (uint32_t l = 0; l <= NXS_MYERS_MAXPAT; l++) {
		len_off[l + 1] += len_off[l];
	}
	{
		uint32_t next[NXS_MYERS_MAXPAT + 2];
		memcpy(next, len_off, sizeof(next));
		for (uint32_t i = 0; i < n_tok; i++) {
			rank[i] = next[roff[i + 1] - roff[i]]++;
		}
	}
	memcpy(sl.pin + up_bytes, tok_bytes + tok_off[0], blen);
	FZT(0);
	if (hipMemcpyAsync(d_upb, sl.pin, up_bytes + blen, hipMemcpyHostToDevice, st) != hipSuccess) {
		set_error("fuzzy upload failed");
		return -1;
	}
	FZT(1);
	if (hipMemsetAsync(d_cnt, 0, 16 + 16 + FZ_NQ * FZ_CSTRIDE * 4, st) != hipSuccess ||
	    hipMemsetAsync(d_tokf + n_tok, 0xff, 4 * sizeof(uint2), st) != hipSuccess) {
		set_error("fuzzy upload failed");
		return -1;
	}
	FZT(2);
	if (ix->profiling) (void)hipEventRecord(sl.ev[0], st);
	hipLaunchKernelGGL(k_bk_peq, dim3(n_tok), dim3(256), 0, st, d_bytes, d_off, n_tok, d_peq, d_tokf, d_rank);
	hipLaunchKernelGGL(k_bk_seed, dim3((n_tok + 255) / 256), dim3(256), 0, st, d_cand, d_cnt + 3, n_tok, d_best,
	    (unsigned long long *)NULL);
	if (n_c) {
		hipLaunchKernelGGL(k_fz_filter, dim3((n_c + 255) / 256, gy), dim3(256), 0, st,
		    ix->d_fz_sig, ix->d_fz_node, ix->d_fz_len, n_c, d_tokf, d_len_off, d_cand, d_qcnt,
		    (uint32_t)qcap, d_cnt + 2);
	}
	if (ix->profiling) (void)hipEventRecord(sl.ev[2], st);
	memset(&fa, 0, sizeof(fa));
	fa.bk = ix->d_bk;
	fa.bk_bytes = ix->d_bk_bytes;
	fa.tok_bytes = d_bytes;
	fa.tok_off = d_off;
	fa.peq = d_peq;
	fa.cap = (uint32_t)std::min<uint64_t>(ccap, 0xffffffffu);
	fa.best = d_best;
	fa.overflow = d_cnt + 2;
	fa.prune = 1;
	fa.evals = ix->profiling ? d_evals : NULL;
	hipLaunchKernelGGL(k_fz_dist, dim3(8, FZ_NQ), dim3(1024), 0, st, fa, d_cand, d_qcnt, (uint32_t)qcap, d_match, d_cnt + 1,
	    (uint32_t)std::min<uint64_t>(mcap, 0xffffffffu));
	if (ix->profiling) (void)hipEventRecord(sl.ev[3], st);
	hipLaunchKernelGGL(k_fz_chain, dim3(1024), dim3(256), 0, st, fa, ix->d_bk_parent, ix->d_bk_slot, d_match, d_cnt + 1,
	    (uint32_t)std::min<uint64_t>(mcap, 0xffffffffu));
	hipLaunchKernelGGL(k_bk_finish, dim3((n_tok + 255) / 256), dim3(256), 0, st, ix->d_bk, d_best, n_tok, d_tids);
	if (ix->profiling) (void)hipEventRecord(sl.ev[1], st);
	if (hipGetLastError() != hipSuccess) {
		set_error("fuzzy kernel launch failed");
		return -1;
	}
	FZT(3);
#ifdef NXS_DBG_FZT
	if (++ncall % 200 == 0) fprintf(stderr, "fzt host %.1f up %.1f memset %.1f kernels %.1f us (prof %d)\n", 1e6*acc[0]/ncall, 1e6*acc[1]/ncall, 1e6*acc[2]/ncall, 1e6*acc[3]/ncall, (int)ix->profiling);
#endif
	/* (one copy into PINNED memory: a copy to pageable memory makes the call wait for the stream) */
	if (hipMemcpyAsync(h_dn, d_dn, dn_bytes + (ix->profiling ? qcnt_bytes : 0), hipMemcpyDeviceToHost, st) != hipSuccess) {
		set_error("fuzzy pass failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}
	/* (the pass's own end: another pass may be queued behind it on the stream before this one is waited for) */
	if (hipEventRecord(sl.ev_done, st) != hipSuccess) {
		set_error("fuzzy pass failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}
	return 0;
}

static int
mf_finish(nxsgpu_index_t *ix, int slot, uint32_t n_tok, uint32_t *term_ids)
{
	nxsgpu_index::fz_slot_t &sl = ix->fzs[slot];
	/* (the staging layout of mf_launch) */
	const uint32_t n_c = ix->n_fz;
	const size_t up_words = (size_t)n_tok + 1 + n_tok + NXS_MYERS_MAXPAT + 2;
	const size_t up_bytes = (up_words * 4 + 15) & ~(size_t)15;
	const uint32_t *const len_off = (const uint32_t *)sl.pin + n_tok + 1 + n_tok;
	const uint32_t blen = ((const uint32_t *)sl.pin)[n_tok];
	const uint8_t *const h_dn = sl.pin + up_bytes + ((blen + 16 + 15) & ~(size_t)15);
	const size_t dn_bytes = ((size_t)n_tok * 4 + 15 & ~(size_t)15) + 16 + 16;
	const uint32_t *const h_qcnt = (const uint32_t *)(h_dn + dn_bytes);	/* (valid when profiling) */

	if (hipEventSynchronize(sl.ev_done) != hipSuccess) {
		set_error("fuzzy pass failed: %s", hipGetErrorString(hipGetLastError()));
		return -1;
	}
	const uint32_t *const h_cnt = (const uint32_t *)(h_dn + (((size_t)n_tok * 4 + 15) & ~(size_t)15));
	const unsigned long long *const h_evals = (const unsigned long long *)(h_cnt + 4);
	memcpy(term_ids, h_dn, (size_t)n_tok * 4);
	if (ix->profiling) {
		float ms = 0;
		(void)hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]);
		ix->prof.fuzzy_ms += ms;
		(void)hipEventElapsedTime(&ms, sl.ev[0], sl.ev[2]);
		ix->prof.fuzzy_filter_ms += ms;
		(void)hipEventElapsedTime(&ms, sl.ev[2], sl.ev[3]);
		ix->prof.fuzzy_dist_ms += ms;
		(void)hipEventElapsedTime(&ms, sl.ev[3], sl.ev[1]);
		ix->prof.fuzzy_chain_ms += ms;
	}
	if (h_cnt[2]) {
		return 1;
	}
	if (ix->profiling) {
		/* distance evaluations; "pairs" = what the queues carried; levels: pairs
		 * screened, survivors, matches */
		uint64_t surv = 0;
		for (uint32_t q = 0; q < FZ_NQ; q++) {
			surv += h_qcnt[q * FZ_CSTRIDE];
		}
		ix->prof.fuzzy_visits += h_evals[0];
		/* the (token, term) pairs k_fz_filter compared: per workgroup of 256 candidates
		 * (sorted by length) the tokens of compatible length -- the kernel's own bounds */
		if (ix->fz_len_start.size() == FZ_MAXLEN + 2) {
			uint64_t checked = 0;
			uint32_t l_lo = 0, l_hi = 0;
			for (uint32_t first = 0; first < n_c; first += 256) {
				const uint32_t last = std::min(first + 255, n_c - 1);
				while (ix->fz_len_start[l_lo + 1] <= first) l_lo++;
				l_hi = std::max(l_hi, l_lo);
				while (ix->fz_len_start[l_hi + 1] <= last) l_hi++;
				const uint32_t ta = len_off[l_lo > 2 ? l_lo - 2 : 0];
				const uint32_t tb = len_off[std::min<uint32_t>(l_hi + 2, NXS_MYERS_MAXPAT) + 1];
				checked += (uint64_t)(last - first + 1) * (tb > ta ? tb - ta : 0);
			}
			ix->prof.fuzzy_checked += checked;
		}
		ix->prof.fuzzy_pairs += surv + h_cnt[1];
		ix->prof.fuzzy_level[0] += (uint64_t)n_tok * n_c;
		ix->prof.fuzzy_level[1] += surv;
		ix->prof.fuzzy_level[2] += h_cnt[1];
	}
	return 0;
}

/* can the batch take the match-first pass as ONE launch sequence (the usual case: no visit counts wanted,
 * every token fits the bit-vector distance)? */
static bool
mf_whole_batch(const nxsgpu_index_t *ix, const uint32_t *tok_off, uint32_t n_tok)
{
	if (ix->cfg.fuzzy_bfs || ix->cfg.fuzzy_noprune || !ix->d_bk_parent || !ix->n_bk || !n_tok) {
		return false;
	}
	for (uint32_t i = 0; i < n_tok; i++) {
		if (tok_off[i + 1] - tok_off[i] > NXS_MYERS_MAXPAT) {
			return false;
		}
	}
	return true;
}

static int fuzzy_search(nxsgpu_index_t *, const uint8_t *, const uint32_t *, uint32_t, uint32_t *, uint64_t *, bool);

static int
mf_free_slot(const nxsgpu_index_t *ix)
{
	for (int i = 0; i < NXSGPU_FZ_SLOTS; i++) {
		if (!ix->fzs[i].state) {
			return i;
		}
	}
	return -1;
}

extern "C" int
nxsgpu_fuzzy(nxsgpu_index_t *ix, const uint8_t *tok_bytes, const uint32_t *tok_off,
    uint32_t n_tok, uint32_t *term_ids, uint64_t *visited)
{
	for (int i = 0; i < NXSGPU_FZ_SLOTS; i++) {
		if (ix->fzs[i].state) {
			set_error("a fuzzy pass is in flight (nxsgpu_fuzzy_begin without _end)");
			return -1;
		}
	}
	return fuzzy_search(ix, tok_bytes, tok_off, n_tok, term_ids, visited, false);
}

/*
 * The same search in two calls: _begin queues the match-first pass of the whole batch on the fuzzy stream
 * and returns a slot (>= 0; -1: error, -2: both slots are taken); _end (slot, same tokens) waits for it
 * and delivers the term ids -- or runs the level-by-level search when a queue of the pass overflowed, and
 * the whole search when the batch did not qualify for a single pass.  NXSGPU_FZ_SLOTS passes can be in
 * flight, each with workspaces of its own; they run in the order they were begun and are to be ended in
 * that order.  nxsgpu_fuzzy() refuses to run meanwhile.
 */
extern "C" int
nxsgpu_fuzzy_begin(nxsgpu_index_t *ix, const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tok)
{
	const int slot = mf_free_slot(ix);

	if (slot < 0) {
		set_error("%d fuzzy passes are in flight already", NXSGPU_FZ_SLOTS);
		return -2;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	ix->fzs[slot].n = n_tok;
	if (!mf_whole_batch(ix, tok_off, n_tok)) {
		ix->fzs[slot].state = 2;
		return slot;
	}
	if (mf_launch(ix, slot, tok_bytes, tok_off, n_tok) != 0) {
		(void)hipStreamSynchronize(ix->stream_fz);
		return -1;
	}
	ix->fzs[slot].state = 1;
	return slot;
}

extern "C" int
nxsgpu_fuzzy_end(nxsgpu_index_t *ix, int slot, const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tok,
    uint32_t *term_ids)
{
	if (slot < 0 || slot >= NXSGPU_FZ_SLOTS || !ix->fzs[slot].state || n_tok != ix->fzs[slot].n) {
		set_error("nxsgpu_fuzzy_end without a matching _begin");
		return -1;
	}
	const int mode = ix->fzs[slot].state;
	if (hipSetDevice(ix->device) != hipSuccess) {
		ix->fzs[slot].state = 0;
		set_error("hipSetDevice failed");
		return -1;
	}
	if (mode == 1) {
		const int r = mf_finish(ix, slot, n_tok, term_ids);
		ix->fzs[slot].state = 0;
		if (r <= 0) {
			return r;
		}
		return fuzzy_search(ix, tok_bytes, tok_off, n_tok, term_ids, NULL, true);
	}
	ix->fzs[slot].state = 0;
	return fuzzy_search(ix, tok_bytes, tok_off, n_tok, term_ids, NULL, false);
}

/* (no_mf: the match-first pass of these tokens has been tried and overflowed) */
static int
fuzzy_search(nxsgpu_index_t *ix, const uint8_t *tok_bytes, const uint32_t *tok_off,
    uint32_t n_tok, uint32_t *term_ids, uint64_t *visited, bool no_mf)
{
	const uint64_t budget = ix->cfg.fuzzy_items;
	const uint32_t n_bk = ix->n_bk;
	uint32_t chunk, max_len = 0;
	bool any_long = false;

	if (n_tok == 0) {
		return 0;
	}
	if (n_bk == 0) {
		memset(term_ids, 0, n_tok * sizeof(uint32_t));
		if (visited) memset(visited, 0, n_tok * sizeof(uint64_t));
		return 0;
	}
	if (hipSetDevice(ix->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	for (uint32_t i = 0; i < n_tok; i++) {
		const uint32_t m = tok_off[i + 1] - tok_off[i];
		max_len = std::max(max_len, m);
		if (m > NXS_MYERS_MAXPAT) {
			any_long = true;
		}
	}
	/*
	 * Worst case one token visits every node: with `safe_chunk` tokens per pass
	 * the frontier queues can never overflow.  A d <= 2 search visits ~10 % of a
	 * large tree, though, so the whole batch is tried in ONE pass first (29
	 * level launches instead of 29 per chunk, and fuller levels); a pass that
	 * does overflow the queues is repeated with a quarter of the tokens, down to
	 * the safe size.
	 */
	/* the usual case: no visit counts wanted, every token fits the bit-vector
	 * distance -- match first, then reachability; the frontier search below is
	 * what the reference does, step for step, and the fallback */
	if (!visited && !no_mf && !ix->cfg.fuzzy_bfs && !ix->cfg.fuzzy_noprune && ix->d_bk_parent) {
		if (!any_long) {
			/* (a free slot: nxsgpu_fuzzy() runs with none taken, nxsgpu_fuzzy_end() has given its own back) */
			const int slot = mf_free_slot(ix);
			int r = slot < 0 ? 1 : mf_launch(ix, slot, tok_bytes, tok_off, n_tok);
			if (slot >= 0 && r == 0) {
				r = mf_finish(ix, slot, n_tok, term_ids);
			} else if (slot >= 0) {
				(void)hipStreamSynchronize(ix->stream_fz);
			}
			if (r <= 0) {
				return r;
			}
		} else if (!ix->fz_split) {
			/* tokens beyond the bit-vector distance (> 64 bytes) take the frontier
			 * search with its row DP, the others the match-first search */
			std::vector<uint32_t> sel[2], off[2], ids[2];
			std::vector<uint8_t> bytes[2];
			int rc = 0;
			for (uint32_t i = 0; i < n_tok; i++) {
				const uint32_t m = tok_off[i + 1] - tok_off[i];
				const int w = m > NXS_MYERS_MAXPAT;
				if (sel[w].empty()) {
					off[w].push_back(0);
				}
				sel[w].push_back(i);
				bytes[w].insert(bytes[w].end(), tok_bytes + tok_off[i], tok_bytes + tok_off[i + 1]);
				off[w].push_back((uint32_t)bytes[w].size());
			}
			ix->fz_split = true;
			for (int w = 0; w < 2 && rc == 0; w++) {
				if (!sel[w].empty()) {
					ids[w].resize(sel[w].size());
					bytes[w].resize(bytes[w].size() + 16);
					rc = fuzzy_search(ix, bytes[w].data(), off[w].data(), (uint32_t)sel[w].size(), ids[w].data(), NULL, false);
					for (size_t j = 0; j < sel[w].size(); j++) {
						term_ids[sel[w][j]] = ids[w][j];
					}
				}
			}
			ix->fz_split = false;
			return rc;
		}
	}
	const uint32_t safe_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tok, budget / n_bk));
	chunk = ix->cfg.fuzzy_safe ? safe_chunk : n_tok;

	const uint32_t LONG_THREADS = 64 * 64;
	const uint64_t cap = std::max<uint64_t>((uint64_t)safe_chunk * n_bk, std::min<uint64_t>(budget, (uint64_t)chunk * n_bk));
	const size_t levels = (size_t)ix->bk_depth + 2;
	size_t need = 4096 + cap * sizeof(fz_item_t) * 2 + levels * 4 + 256
	    + (size_t)chunk * (256 * 8 + 4 + 8 + 4) + tok_off[n_tok] + 16 + ((size_t)chunk + 1) * 4 + 512
	    + (any_long ? (size_t)LONG_THREADS * ((size_t)max_len + 2) * 2 : 0) + 16 * 256;
	if (ix->fz_len < need) {
		if (ix->fz) {
			(void)hipFree(ix->fz);
			ix->fz = NULL;
			ix->fz_len = 0;
		}
		if (hipMalloc(&ix->fz, need) != hipSuccess) {
			set_error("hipMalloc(%zu) for the fuzzy workspace failed", need);
			return -1;
		}
		ix->fz_len = need;
	}

	for (uint32_t c0 = 0; c0 < n_tok; ) {
		const uint32_t nc = std::min(chunk, n_tok - c0);
		const uint32_t boff = tok_off[c0], blen = tok_off[c0 + nc] - boff;
		std::vector<uint32_t> roff(nc + 1);
		uint8_t *p = (uint8_t *)ix->fz;
		fz_item_t *qa = carve<fz_item_t>(p, cap);
		fz_item_t *qb = carve<fz_item_t>(p, cap);
		uint32_t *counts = carve<uint32_t>(p, levels);
		uint32_t *d_ovf = carve<uint32_t>(p, 1);
		uint64_t *d_peq = carve<uint64_t>(p, (size_t)nc * 256);
		uint32_t *d_best = carve<uint32_t>(p, nc);
		unsigned long long *d_vis = carve<unsigned long long>(p, nc);
		unsigned long long *d_evals = carve<unsigned long long>(p, 1);
		uint32_t *d_tids = carve<uint32_t>(p, nc);
		uint8_t *d_bytes = carve<uint8_t>(p, blen + 16);
		uint32_t *d_off = carve<uint32_t>(p, nc + 1);
		uint16_t *d_rows = any_long ? carve<uint16_t>(p, (size_t)LONG_THREADS * (max_len + 2)) : NULL;
		fz_args_t fa;
		uint32_t h_ovf = 0;

		for (uint32_t i = 0; i <= nc; i++) {
			roff[i] = tok_off[c0 + i] - boff;
		}
		if (hipMemcpyAsync(d_bytes, tok_bytes + boff, blen, hipMemcpyHostToDevice, ix->stream_fz) != hipSuccess ||
		    hipMemcpyAsync(d_off, roff.data(), (nc + 1) * 4, hipMemcpyHostToDevice, ix->stream_fz) != hipSuccess ||
		    hipMemsetAsync(counts, 0, levels * 4, ix->stream_fz) != hipSuccess ||
		    hipMemsetAsync(d_ovf, 0, 4, ix->stream_fz) != hipSuccess ||
		    hipMemsetAsync(d_evals, 0, 8, ix->stream_fz) != hipSuccess) {
			set_error("fuzzy upload failed");
			return -1;
		}
		if (ix->profiling) (void)hipEventRecord(ix->ev[0], ix->stream_fz);
		hipLaunchKernelGGL(k_bk_peq, dim3(nc), dim3(256), 0, ix->stream_fz, d_bytes, d_off, nc, d_peq, (uint2 *)NULL, (const uint32_t *)NULL);
		hipLaunchKernelGGL(k_bk_seed, dim3((nc + 255) / 256), dim3(256), 0, ix->stream_fz,
		    qa, counts, nc, d_best, visited ? d_vis : (unsigned long long *)NULL);

		memset(&fa, 0, sizeof(fa));
		fa.bk = ix->d_bk;
		fa.bk_bytes = ix->d_bk_bytes;
		fa.tok_bytes = d_bytes;
		fa.tok_off = d_off;
		fa.peq = d_peq;
		fa.cap = (uint32_t)std::min<uint64_t>(cap, 0xffffffffu);
		fa.best = d_best;
		fa.visited = visited ? d_vis : NULL;
		fa.dp_rows = d_rows;
		fa.dp_stride = max_len + 2;
		fa.overflow = d_ovf;
		fa.prune = (!visited && !ix->cfg.fuzzy_noprune) ? 1u : 0u;
		fa.evals = ix->profiling ? d_evals : NULL;
		for (uint32_t lvl = 0; lvl < ix->bk_depth; lvl++) {
			fa.cur = (lvl & 1) ? qb : qa;
			fa.next = (lvl & 1) ? qa : qb;
			fa.cur_count = counts + lvl;
			fa.next_count = counts + lvl + 1;
			hipLaunchKernelGGL(k_bk_level<false>, dim3(512), dim3(1024), 0, ix->stream_fz, fa);
			if (any_long) {
				/* tokens longer than 64 bytes: row DP, bounded scratch */
				hipLaunchKernelGGL(k_bk_level<true>, dim3(LONG_THREADS / 64), dim3(64), 0, ix->stream_fz, fa);
			}
		}
		hipLaunchKernelGGL(k_bk_finish, dim3((nc + 255) / 256), dim3(256), 0, ix->stream_fz,
		    ix->d_bk, d_best, nc, d_tids);
		if (ix->profiling) (void)hipEventRecord(ix->ev[1], ix->stream_fz);
		if (hipGetLastError() != hipSuccess) {
			set_error("fuzzy kernel launch failed");
			return -1;
		}
		std::vector<uint32_t> h_counts(levels);
		unsigned long long h_evals = 0;
		if (hipMemcpyAsync(term_ids + c0, d_tids, nc * 4, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    hipMemcpyAsync(&h_evals, d_evals, 8, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    (visited && hipMemcpyAsync(visited + c0, d_vis, nc * 8, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess) ||
		    hipMemcpyAsync(&h_ovf, d_ovf, 4, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    hipMemcpyAsync(h_counts.data(), counts, levels * 4, hipMemcpyDeviceToHost, ix->stream_fz) != hipSuccess ||
		    hipStreamSynchronize(ix->stream_fz) != hipSuccess) {
			set_error("fuzzy pass failed: %s", hipGetErrorString(hipGetLastError()));
			return -1;
		}
		if (ix->profiling) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, ix->ev[0], ix->ev[1]);
			ix->prof.fuzzy_ms += ms;		/* a repeated pass is time spent too */
		}
		if (h_ovf) {
			if (chunk <= safe_chunk) {
				set_error("fuzzy frontier overflow (internal error)");
				return -1;
			}
			chunk = std::max(safe_chunk, chunk / 4);
			continue;		/* same tokens again, fewer at a time */
		}
		if (ix->profiling) {
			/* distance evaluations; (token, node) pairs dequeued, pruned ones
			 * included, are the level counts */
			ix->prof.fuzzy_visits += h_evals;
			for (size_t l = 0; l < levels; l++) {
				ix->prof.fuzzy_pairs += h_counts[l];
				if (l < 40) {
					ix->prof.fuzzy_level[l] += h_counts[l];
				}
			}
		}
		c0 += nc;
	}
	return 0;
}
