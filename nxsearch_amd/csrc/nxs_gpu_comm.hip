/*
 * nxs_gpu_comm.hip -- query sharding: the RCCL communicator (dlopen) and the all-gather of the record blocks
 * (MI355X / gfx950 query path of nxsearch; see nxs_gpu_int.h for the map of the files)
 */
#include "nxs_gpu_int.h"
#include <dlfcn.h>
#include <rccl/rccl.h>		/* types only: the library is dlopen()ed (rccl_api) */

/*
 * RCCL is loaded at first use (dlopen), so that a single-GPU consumer has no
 * link-time dependency on it and a process that already holds an RCCL (PyTorch
 * ships its own copy) keeps using that one.
 */
struct rccl_api_t {
	void *		handle;
	ncclResult_t	(*GetUniqueId)(ncclUniqueId *);
	ncclResult_t	(*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
	ncclResult_t	(*CommDestroy)(ncclComm_t);
	ncclResult_t	(*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
	ncclResult_t	(*CommCount)(const ncclComm_t, int *);		/* (optional: evidence only) */
	const char *	(*GetErrorString)(ncclResult_t);
};

static rccl_api_t *
rccl_api(void)
{
	static rccl_api_t api;
	static bool tried = false;

	if (tried) {
		return api.handle ? &api : NULL;
	}
	tried = true;
	static const char *const names[] = { "librccl.so.1", "librccl.so" };
	void *h = NULL;
	for (int pass = 0; pass < 2 && !h; pass++) {
		for (size_t i = 0; i < 2 && !h; i++) {
			h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
		}
	}
	if (!h) {
		h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
	}
	if (!h) {
		set_error("cannot load librccl: %s", dlerror());
		return NULL;
	}
	api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
	api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
	api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
	api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
	api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
	api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
	if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
		set_error("librccl lacks an expected symbol");
		return NULL;
	}
	api.handle = h;
	return &api;
}

struct nxsgpu_comm {
	int		device, rank, world;
	ncclComm_t	comm;
	hipStream_t	stream;		/* blocking helper's own stream */
	void *		d_buf;
	size_t		d_len;
	uint64_t	n_gathers, gather_bytes;	/* all-gathers queued so far, bytes this rank contributed */
};

static const char *
rccl_err(rccl_api_t *R, ncclResult_t r)
{
	return R->GetErrorString ? R->GetErrorString(r) : "rccl error";
}

extern "C" int
nxsgpu_comm_unique_id(uint8_t uid[NXSGPU_UID_BYTES])
{
	rccl_api_t *R = rccl_api();
	ncclUniqueId id;
	ncclResult_t r;

	static_assert(sizeof(ncclUniqueId) == NXSGPU_UID_BYTES, "uid size");
	if (!R) {
		return -1;
	}
	if ((r = R->GetUniqueId(&id)) != ncclSuccess) {
		set_error("ncclGetUniqueId: %s", rccl_err(R, r));
		return -1;
	}
	memcpy(uid, &id, NXSGPU_UID_BYTES);
	return 0;
}

extern "C" nxsgpu_comm_t *
nxsgpu_comm_create(int device, int rank, int world, const uint8_t uid[NXSGPU_UID_BYTES])
{
	rccl_api_t *R = rccl_api();
	nxsgpu_comm_t *c;
	ncclUniqueId id;
	ncclResult_t r;

	if (!R) {
		return NULL;
	}
	if (world < 1 || rank < 0 || rank >= world) {
		set_error("bad rank %d of %d", rank, world);
		return NULL;
	}
	if (hipSetDevice(device) != hipSuccess) {
		set_error("hipSetDevice(%d) failed", device);
		return NULL;
	}
	c = new nxsgpu_comm();
	memset(c, 0, sizeof(*c));
	c->device = device;
	c->rank = rank;
	c->world = world;
	memcpy(&id, uid, NXSGPU_UID_BYTES);
	if ((r = R->CommInitRank(&c->comm, world, id, rank)) != ncclSuccess) {
		set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(R, r));
		delete c;
		return NULL;
	}
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
		set_error("hipStreamCreate failed");
		(void)R->CommDestroy(c->comm);
		delete c;
		return NULL;
	}
	return c;
}

extern "C" void
nxsgpu_comm_destroy(nxsgpu_comm_t *c)
{
	rccl_api_t *R = rccl_api();

	if (!c) {
		return;
	}
	(void)hipSetDevice(c->device);
	if (c->stream) {
		(void)hipStreamSynchronize(c->stream);
		(void)hipStreamDestroy(c->stream);
	}
	(void)hipFree(c->d_buf);
	if (R && c->comm) {
		(void)R->CommDestroy(c->comm);
	}
	delete c;
}

extern "C" int nxsgpu_comm_rank(const nxsgpu_comm_t *c) { return c ? c->rank : 0; }
extern "C" int nxsgpu_comm_world(const nxsgpu_comm_t *c) { return c ? c->world : 1; }

/* what RCCL itself says the communicator spans (ncclCommCount): -1 if it cannot be asked */
extern "C" int
nxsgpu_comm_rccl_count(const nxsgpu_comm_t *c)
{
	rccl_api_t *R = rccl_api();
	int n = -1;

	if (!R || !c || !R->CommCount || R->CommCount(c->comm, &n) != ncclSuccess) {
		return -1;
	}
	return n;
}

/* out[0] = all-gathers queued on this communicator so far, out[1] = bytes this rank contributed to them */
extern "C" void
nxsgpu_comm_stats(const nxsgpu_comm_t *c, uint64_t out[2])
{
	out[0] = c ? c->n_gathers : 0;
	out[1] = c ? c->gather_bytes : 0;
}

/* device buffers, asynchronous on `stream`; recv holds world x bytes */
int
comm_allgather_dev(nxsgpu_comm_t *c, const void *send, void *recv, size_t bytes, hipStream_t stream)
{
	rccl_api_t *R = rccl_api();
	ncclResult_t r;

	if (!R || !c) {
		set_error("no communicator");
		return -1;
	}
	if ((r = R->AllGather(send, recv, bytes, ncclChar, c->comm, stream)) != ncclSuccess) {
		set_error("ncclAllGather: %s", rccl_err(R, r));
		return -1;
	}
	c->n_gathers++;
	c->gather_bytes += bytes;
	return 0;
}

extern "C" int
nxsgpu_comm_allgather(nxsgpu_comm_t *c, const void *send, void *recv, size_t bytes)
{
	const size_t need = (size_t)(c->world + 1) * bytes + 512;
	uint8_t *d_send, *d_recv;

	if (hipSetDevice(c->device) != hipSuccess) {
		set_error("hipSetDevice failed");
		return -1;
	}
	if (bytes == 0) {
		return 0;
	}
	if (c->d_len < need) {
		(void)hipFree(c->d_buf);
		c->d_buf = NULL;
		c->d_len = 0;
		if (hipMalloc(&c->d_buf, need) != hipSuccess) {
			set_error("hipMalloc(%zu) failed", need);
			return -1;
		}
		c->d_len = need;
	}
	d_send = (uint8_t *)c->d_buf;
	d_recv = d_send + ((bytes + 255) & ~(size_t)255);
	clear_error();
	if (hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
	    comm_allgather_dev(c, d_send, d_recv, bytes, c->stream) != 0 ||
	    hipMemcpyAsync(recv, d_recv, (size_t)c->world * bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
	    hipStreamSynchronize(c->stream) != hipSuccess) {
		if (!have_error()) {
			set_error("all-gather failed");
		}
		return -1;
	}
	return 0;
}

extern "C" int
nxsgpu_index_set_comm(nxsgpu_index_t *ix, nxsgpu_comm_t *c)
{
	if (nxsgpu_batches_in_flight(ix)) {
		set_error("nxsgpu_index_set_comm: batches are in flight");
		return -1;
	}
	if (c && c->device != ix->device) {
		set_error("communicator and index live on different devices (%d, %d)", c->device, ix->device);
		return -1;
	}
	ix->comm = c;
	return 0;
}

