/* rccl_stub.c -- TEST INFRASTRUCTURE ONLY.  The four nccl* entry points libfastdepth binds at run time (csrc/fd_train_bwd_impl.h: RcclApi), implemented
 * for HOST memory over a POSIX shared-memory segment, so that tests/test_dp_gloo.py can drive the library-issued gradient exchange
 * (fd_train_backward_allreduce) of the CPU-emulator build at world size > 1.  Synchronous: "streams" are ignored (the emulator's launches are
 * synchronous too).  Sum semantics follow RCCL's ring for the cases the tests check: fp32 sums in rank order (commutative for 2 ranks, so bit-equal to
 * any order), bfloat16 sums in fp32 rounded once to bfloat16 (round to nearest even).
 *   build: gcc -O2 -shared -fPIC -o librccl_stub.so rccl_stub.c -lrt -pthread   (tests/rccl_stub/build_stub.py)                               */
#define _GNU_SOURCE
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define ID_BYTES 128
#define SLOT_BYTES (4u << 20)          /* per rank and exchange round */
#define MAX_RANKS 8
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 };
enum { ncclFloat32 = 7, ncclBfloat16 = 9 };
typedef struct { char internal[ID_BYTES]; } ncclUniqueId;

typedef struct {
    atomic_int ready;                  /* 1 once the creator has initialised the header */
    atomic_int arrived;                /* barrier: arrivals of the current generation */
    atomic_int generation;
    atomic_int attached, detached;
} header_t;
typedef struct { header_t *h; unsigned char *slots; int rank, nranks; size_t bytes; char name[64]; } comm_t;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int barrier(comm_t *c)
{
    const int gen = atomic_load(&c->h->generation);
    if (atomic_fetch_add(&c->h->arrived, 1) == c->nranks - 1) {
        atomic_store(&c->h->arrived, 0);
        atomic_fetch_add(&c->h->generation, 1);
        return 0;
    }
    const double t0 = now_s();
    while (atomic_load(&c->h->generation) == gen) {
        sched_yield();
        if (now_s() - t0 > 120.0) return -1;        /* a peer died: fail instead of hanging the test tier */
    }
    return 0;
}

int ncclGetUniqueId(ncclUniqueId *id)
{
    static atomic_int counter;
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, 64, "/fd_rccl_stub_%d_%d_%ld", (int)getpid(), atomic_fetch_add(&counter, 1), (long)time(NULL));
    return ncclSuccess;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks <= 0 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    comm_t *c = (comm_t *)calloc(1, sizeof(comm_t));
    c->rank = rank; c->nranks = nranks;
    c->bytes = 4096 + (size_t)nranks * SLOT_BYTES;
    memcpy(c->name, id.internal, 63);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); free(c); return ncclSystemError; }
    void *m = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { free(c); return ncclSystemError; }
    c->h = (header_t *)m; c->slots = (unsigned char *)m + 4096;      /* a fresh segment is zero-filled: arrived = generation = 0 */
    atomic_fetch_add(&c->h->attached, 1);
    if (barrier(c) != 0) { munmap(m, c->bytes); free(c); return ncclSystemError; }     /* everyone is attached before anyone exchanges */
    *comm = c;
    return ncclSuccess;
}

static float bf16_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f32_to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, void *stream)
{
    (void)stream;
    comm_t *c = (comm_t *)comm;
    if (!c || !send || !recv || op != 0 || (dtype != ncclFloat32 && dtype != ncclBfloat16)) return ncclInvalidArgument;
    const size_t esz = dtype == ncclFloat32 ? 4 : 2, per = SLOT_BYTES / esz;
    for (size_t done = 0; done < count; done += per) {
        const size_t n = count - done < per ? count - done : per;
        memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, (const unsigned char *)send + done * esz, n * esz);
        if (barrier(c) != 0) return ncclSystemError;
        if (dtype == ncclFloat32) {
            float *out = (float *)recv + done;
            for (size_t i = 0; i < n; ++i) {
                float s = ((const float *)c->slots)[i];
                for (int r = 1; r < c->nranks; ++r) s += ((const float *)(c->slots + (size_t)r * SLOT_BYTES))[i];
                out[i] = s;
            }
        } else {
            uint16_t *out = (uint16_t *)recv + done;
            for (size_t i = 0; i < n; ++i) {
                float s = bf16_to_f32(((const uint16_t *)c->slots)[i]);
                for (int r = 1; r < c->nranks; ++r) s += bf16_to_f32(((const uint16_t *)(c->slots + (size_t)r * SLOT_BYTES))[i]);
                out[i] = f32_to_bf16(s);
            }
        }
        if (barrier(c) != 0) return ncclSystemError;        /* every rank has read the slots before the next round overwrites them */
    }
    return ncclSuccess;
}

int ncclCommDestroy(void *comm)
{
    comm_t *c = (comm_t *)comm;
    if (!c) return ncclInvalidArgument;
    const int last = atomic_fetch_add(&c->h->detached, 1) == c->nranks - 1;
    munmap((void *)c->h, c->bytes);
    if (last) shm_unlink(c->name);
    free(c);
    return ncclSuccess;
}

const char *ncclGetErrorString(int rc) { return rc == ncclSuccess ? "no error" : (rc == ncclSystemError ? "shared-memory / barrier failure (rccl_stub)" : "invalid argument (rccl_stub)"); }
