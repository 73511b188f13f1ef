// TEST INFRASTRUCTURE (never part of the product): a stand-in for the handful of RCCL entry points libgysketch.so calls, so that the
// in-library window exchange (gys_window_close_rccl, gys_tdigest_global_rccl) can run with MORE THAN ONE RANK on a box with a single
// GPU -- RCCL itself refuses two ranks on one device.  Named by $GYS_RCCL_LIB in the rank processes of tests/test_gpu_round3.py (the
// library binds RCCL with dlopen, gys_engine.hip: rccl_api); the ranks meet
// in a POSIX shared-memory segment named after the unique id.  Every collective is executed for real (device -> shared memory ->
// reduce / gather in rank order -> device) with the count, datatype, operator and pointers the library passed, so section dtype / op
// mapping, buffer placement of the all-gather and the rank-order fold are exercised exactly as a real communicator would see them.
// What it does NOT test is RCCL itself (transport, xGMI, stream semantics beyond "in order on the given stream").
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {
constexpr size_t SLOT = 32u << 20; // bytes per rank and step
constexpr int MAXR = 8;
struct Shared {
	std::atomic<uint32_t> arrived, gen, joined;
	uint32_t pad;
};
struct Comm {
	int rank, nranks;
	Shared *sh;
	uint8_t *slots; // [nranks][SLOT]
	size_t map_bytes;
	char name[64];
	uint64_t ncalls;
};
int g_group = 0;
uint64_t g_allreduce = 0, g_allgather = 0;

void barrier(Comm *c)
{
	const uint32_t g = c->sh->gen.load();
	if (c->sh->arrived.fetch_add(1) + 1 == (uint32_t)c->nranks) {
		c->sh->arrived.store(0);
		c->sh->gen.fetch_add(1);
	} else {
		struct timespec t0;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		while (c->sh->gen.load() == g) {
			sched_yield();
			struct timespec t1;
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if (t1.tv_sec - t0.tv_sec > 120) {
				fprintf(stderr, "fakerccl: rank %d waited 120 s for the other ranks\n", c->rank);
				abort();
			}
		}
	}
}

size_t dt_size(ncclDataType_t t)
{
	switch (t) {
	case ncclInt8: case ncclUint8: return 1;
	case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
	case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
	default: return 0;
	}
}

template <typename T>
void reduce_t(T *acc, const T *in, size_t n, ncclRedOp_t op)
{
	for (size_t i = 0; i < n; ++i) {
		if (op == ncclSum) acc[i] = (T)(acc[i] + in[i]);
		else if (op == ncclMax) acc[i] = acc[i] > in[i] ? acc[i] : in[i];
		else if (op == ncclMin) acc[i] = acc[i] < in[i] ? acc[i] : in[i];
	}
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	memset(id, 0, sizeof(*id));
	struct timespec t;
	clock_gettime(CLOCK_REALTIME, &t);
	snprintf(id->internal, sizeof(id->internal), "gysfake_%d_%lld_%ld", (int)getpid(), (long long)t.tv_sec, t.tv_nsec);
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
	if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
	Comm *c = new Comm();
	c->rank = rank;
	c->nranks = nranks;
	snprintf(c->name, sizeof(c->name), "/%.60s", id.internal);
	c->map_bytes = 4096 + (size_t)nranks * SLOT;
	const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
	void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (m == MAP_FAILED) return ncclSystemError;
	c->sh = (Shared *)m; // (a fresh segment is zero-filled: counters start at 0)
	c->slots = (uint8_t *)m + 4096;
	c->sh->joined.fetch_add(1);
	struct timespec t0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	while (c->sh->joined.load() < (uint32_t)nranks) { // collective, like the real call
		sched_yield();
		struct timespec t1;
		clock_gettime(CLOCK_MONOTONIC, &t1);
		if (t1.tv_sec - t0.tv_sec > 120) return ncclSystemError;
	}
	*comm = (ncclComm_t)c;
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	Comm *c = (Comm *)comm;
	barrier(c);
	if (c->rank == 0) shm_unlink(c->name);
	munmap((void *)c->sh, c->map_bytes);
	delete c;
	return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { *count = ((Comm *)comm)->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) { *rank = ((Comm *)comm)->rank; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fakerccl error"; }
ncclResult_t ncclGroupStart() { ++g_group; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return g_group-- > 0 ? ncclSuccess : ncclInvalidUsage; }

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
	Comm *c = (Comm *)comm;
	const size_t es = dt_size(datatype);
	if (!es || (op != ncclSum && op != ncclMax && op != ncclMin)) return ncclInvalidArgument;
	++g_allreduce;
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
	std::vector<uint8_t> acc;
	for (size_t done = 0; done < count;) {
		const size_t n = std::min(count - done, SLOT / es);
		uint8_t *mine = c->slots + (size_t)c->rank * SLOT;
		if (hipMemcpy(mine, (const uint8_t *)sendbuff + done * es, n * es, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
		barrier(c);
		acc.assign(c->slots, c->slots + n * es); // rank 0's contribution, then the others in rank order
		for (int r = 1; r < c->nranks; ++r) {
			const uint8_t *in = c->slots + (size_t)r * SLOT;
			switch (datatype) {
			case ncclUint8: reduce_t((uint8_t *)acc.data(), (const uint8_t *)in, n, op); break;
			case ncclInt8: reduce_t((int8_t *)acc.data(), (const int8_t *)in, n, op); break;
			case ncclUint32: reduce_t((uint32_t *)acc.data(), (const uint32_t *)in, n, op); break;
			case ncclInt32: reduce_t((int32_t *)acc.data(), (const int32_t *)in, n, op); break;
			case ncclUint64: reduce_t((uint64_t *)acc.data(), (const uint64_t *)in, n, op); break;
			case ncclInt64: reduce_t((int64_t *)acc.data(), (const int64_t *)in, n, op); break;
			case ncclFloat32: reduce_t((float *)acc.data(), (const float *)in, n, op); break;
			case ncclFloat64: reduce_t((double *)acc.data(), (const double *)in, n, op); break;
			default: return ncclInvalidArgument;
			}
		}
		barrier(c); // everyone has read the slots before the next step overwrites them
		if (hipMemcpy((uint8_t *)recvbuff + done * es, acc.data(), n * es, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
		done += n;
	}
	return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
	Comm *c = (Comm *)comm;
	const size_t bytes = sendcount * dt_size(datatype);
	if (!dt_size(datatype) || bytes > SLOT) return ncclInvalidArgument;
	++g_allgather;
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
	if (hipMemcpy(c->slots + (size_t)c->rank * SLOT, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
	barrier(c);
	for (int r = 0; r < c->nranks; ++r)
		if (hipMemcpy((uint8_t *)recvbuff + (size_t)r * bytes, c->slots + (size_t)r * SLOT, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
	barrier(c);
	return ncclSuccess;
}

// for the test: proof that the library's calls really came through here
uint64_t fakerccl_allreduce_calls() { return g_allreduce; }
uint64_t fakerccl_allgather_calls() { return g_allgather; }
}
