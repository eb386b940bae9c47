// Test infrastructure, not product code: an nccl-API stand-in that lets N processes sharing ONE GPU exercise libppo_hip's
// data-parallel code path (tests/test_dp_two_ranks.py).  The collectives go through a POSIX shared-memory segment named
// by the unique id: every rank copies its buffer to its slot, a barrier, every rank sums the slots in RANK ORDER (so
// all ranks get bit-identical results, like a ring/tree all-reduce with a fixed schedule), a second barrier.
// Selected with PPO_RCCL_LIBRARY=<this .so>.   Build: hipcc -shared -fPIC -o libfake_rccl.so fake_rccl.cpp -lrt
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {
constexpr size_t SLOT_BYTES = 16u << 20;
struct Header { std::atomic<int> arrived; std::atomic<int> generation; std::atomic<int> attached; };
struct Comm { int world, rank, fd; char name[64]; Header* hdr; char* slots; size_t bytes; std::vector<float> tmp; };

int barrier(Comm* c) {
    const int gen = c->hdr->generation.load(std::memory_order_acquire);
    if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
        c->hdr->arrived.store(0, std::memory_order_relaxed);
        c->hdr->generation.fetch_add(1, std::memory_order_acq_rel);
        return 0;
    }
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    while (c->hdr->generation.load(std::memory_order_acquire) == gen) {
        timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        if (t.tv_sec - t0.tv_sec > 120) { fprintf(stderr, "fake_rccl: barrier timeout (rank %d)\n", c->rank); return 6; }
        usleep(20);
    }
    return 0;
}
}  // namespace

extern "C" {
struct FakeUid { char b[128]; };

int ncclGetUniqueId(FakeUid* uid) {
    memset(uid->b, 0, sizeof uid->b);
    timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf(uid->b, sizeof uid->b, "/ppo_fake_rccl_%d_%ld", (int)getpid(), (long)t.tv_nsec);
    return 0;
}

int ncclCommInitRank(void** comm, int world, FakeUid uid, int rank) {
    Comm* c = new Comm();
    c->world = world; c->rank = rank;
    snprintf(c->name, sizeof c->name, "%s", uid.b);
    c->bytes = 4096 + (size_t)world * SLOT_BYTES;
    c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)c->bytes) != 0) { delete c; return 2; }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (p == MAP_FAILED) { delete c; return 2; }
    c->hdr = (Header*)p;                       // a fresh segment is zero-filled: the atomics start at 0
    c->slots = (char*)p + 4096;
    c->hdr->attached.fetch_add(1);
    *comm = c;
    return barrier(c);
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (dtype != 7 || op != 0) return 4;       // ncclFloat32, ncclSum only
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    const size_t per = SLOT_BYTES / sizeof(float);
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        float* mine = (float*)(c->slots + (size_t)c->rank * SLOT_BYTES);
        if (hipMemcpy(mine, (const float*)send + off, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (int rc = barrier(c)) return rc;
        c->tmp.assign(n, 0.f);
        for (int r = 0; r < c->world; ++r) {
            const float* s = (const float*)(c->slots + (size_t)r * SLOT_BYTES);
            for (size_t i = 0; i < n; ++i) c->tmp[i] += s[i];
        }
        if (int rc = barrier(c)) return rc;
        if (hipMemcpy((float*)recv + off, c->tmp.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (dtype != 7) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    const size_t per = SLOT_BYTES / sizeof(float);
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, (const float*)send + off, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (int rc = barrier(c)) return rc;
        for (int r = 0; r < c->world; ++r)
            if (hipMemcpy((float*)recv + (size_t)r * count + off, c->slots + (size_t)r * SLOT_BYTES, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
        if (int rc = barrier(c)) return rc;
    }
    return 0;
}

int ncclCommCount(void* comm, int* count) { *count = ((Comm*)comm)->world; return 0; }

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    const bool last = c->hdr->attached.fetch_sub(1) == 1;
    munmap((void*)c->hdr, c->bytes);
    close(c->fd);
    if (last) shm_unlink(c->name);
    delete c;
    return 0;
}

const char* ncclGetErrorString(int rc) {
    switch (rc) { case 0: return "success"; case 1: return "fake_rccl: HIP call failed"; case 2: return "fake_rccl: shared memory setup failed";
                  case 4: return "fake_rccl: only float32 sum is implemented"; case 6: return "fake_rccl: barrier timeout"; }
    return "fake_rccl: error";
}
}
