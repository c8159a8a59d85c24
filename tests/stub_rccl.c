/* A stand-in for librccl that runs on the host: the five entry points libosqp_amd.so resolves with dlsym
 * (csrc/comm.hip: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString) plus
 * ncclCommCount, implemented over a shared-memory file between the processes of one machine.  Test infrastructure
 * (tests/test_rccl_stub_transport.py): it lets the library's RCCL transport -- the unique-id hand-over, the rank /
 * world bookkeeping, the in-place all-gather call with its send pointer inside the receive buffer -- run with more
 * than one rank on a box without GPUs.  "Device" pointers are host pointers here; the stream argument is ignored.
 * Layout of the shared file: [64-byte header: arrived, generation] [world x kSlot bytes of payload]. */
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
enum { kSlot = 1 << 20, kHeader = 64 };

typedef struct stub_comm {
  int rank, world, fd;
  volatile int *hdr; /* [0] arrived, [1] generation */
  char *slots;
  size_t bytes;
  char path[120];
} stub_comm;
typedef stub_comm *ncclComm_t;

static int barrier(stub_comm *c) {
  const int gen = __atomic_load_n(&c->hdr[1], __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&c->hdr[0], 1, __ATOMIC_ACQ_REL) == c->world) {
    __atomic_store_n(&c->hdr[0], 0, __ATOMIC_RELEASE);
    __atomic_add_fetch(&c->hdr[1], 1, __ATOMIC_ACQ_REL);
    return 0;
  }
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  while (__atomic_load_n(&c->hdr[1], __ATOMIC_ACQUIRE) == gen) {
    usleep(50);
    clock_gettime(CLOCK_MONOTONIC, &t);
    if (t.tv_sec - t0.tv_sec > 60) return 1; /* a missing rank is an error, not a hang */
  }
  return 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/tmp/stub_rccl_%d_%ld", (int)getpid(), (long)time(NULL));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (getenv("STUB_RCCL_FAIL_INIT")) return ncclSystemError; /* a node whose collective library cannot connect its ranks */
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
  stub_comm *c = (stub_comm *)calloc(1, sizeof(stub_comm));
  c->rank = rank; c->world = nranks;
  memcpy(c->path, id.internal, sizeof(c->path) - 1);
  c->bytes = kHeader + (size_t)nranks * kSlot;
  c->fd = open(c->path, O_RDWR | O_CREAT, 0600);
  if (c->fd < 0 || ftruncate(c->fd, (off_t)c->bytes) != 0) { free(c); return ncclSystemError; }
  void *p = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
  if (p == MAP_FAILED) { close(c->fd); free(c); return ncclSystemError; }
  c->hdr = (volatile int *)p;
  c->slots = (char *)p + kHeader;
  if (barrier(c)) { free(c); return ncclSystemError; } /* as the real call: returns once every rank has joined */
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->world;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t c, void *stream) {
  (void)stream;
  if (!c || dt != ncclDouble || count * 8 > kSlot) return ncclInvalidArgument;
  const size_t bytes = count * 8;
  memcpy(c->slots + (size_t)c->rank * kSlot, send, bytes);
  if (barrier(c)) return ncclSystemError;
  for (int r = 0; r < c->world; r++)
    if ((char *)recv + (size_t)r * bytes != (const char *)send) memcpy((char *)recv + (size_t)r * bytes, c->slots + (size_t)r * kSlot, bytes);
  if (barrier(c)) return ncclSystemError; /* the slots are free for the next call */
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  munmap((void *)c->hdr, c->bytes);
  close(c->fd);
  if (c->rank == 0) unlink(c->path);
  free(c);
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  return r == ncclSuccess ? "no error" : (r == ncclSystemError ? "stub: system error or a rank never arrived" : "stub: invalid argument");
}
