// jolt_amd/csrc/shm_exchange.hip -- the per-round exchange of partial round sums between the ranks of ONE node through POSIX
// shared memory (host code only; no device work in this file).
//
// The round sums of a rank are a few hundred bytes that already sit in host memory when the round's completion flag arrives
// (finish_member writes them there), and every rank needs every other rank's before it can draw the challenge.  Between the
// processes of one node that is a memcpy and a sequence number per rank: ~1 us, against the ~20-30 us of an 8-rank RCCL
// all-gather of the same bytes (kernel launch + ring steps + completion) on the critical path of each of the ~50 exchanged rounds
// of a proof.  RCCL keeps the bulk traffic (the table hand-over to the redundant tail, the MSM partial sums; comm.hip); a rank
// that cannot map the segment falls back to jolt_comm_gather_round_sums (decided collectively by the launcher).
//
// Layout: header, then world x 2 slots {seq, bytes, payload[max_bytes]}.  Exchange number s uses slot [rank][s & 1]: a rank writes
// its payload, then release-stores seq = s, then acquire-spins on the other ranks' seq == s.  Two slots are enough: a rank can
// only be ONE exchange ahead of the slowest (it needs everybody's payload s to finish s), so slot s & 1 is rewritten -- by exchange
// s + 2 -- only after every rank has published s + 1, i.e. has finished reading s.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "../../include/jolt_hip.h"

namespace {
constexpr uint64_t kShmMagic = 0x6a6f6c7473686d31ull;  // "joltshm1"
constexpr double kShmTimeoutSeconds = 60.0;

struct ShmHeader {
    std::atomic<uint64_t> magic;
    uint64_t world, max_bytes, slot_stride;
    uint64_t nonce;  // per-run value chosen by the launcher: tells this run's segment from a stale one of the same name
    char pad[64 - 5 * sizeof(uint64_t)];
};
struct ShmSlot {  // followed by max_bytes of payload; slot_stride keeps every slot on its own cache lines
    std::atomic<uint64_t> seq;
    uint64_t bytes;
    char pad[64 - 2 * sizeof(uint64_t)];
};
static_assert(sizeof(ShmHeader) == 64 && sizeof(ShmSlot) == 64, "shm layout");
}  // namespace

struct jolt_shm {
    std::string name;
    int32_t rank = 0, world = 1;
    size_t max_bytes = 0, map_bytes = 0;
    char* base = nullptr;
    uint64_t seq = 0;
    bool owner = false, unlinked = false;
    ShmHeader* header() const { return reinterpret_cast<ShmHeader*>(base); }
    ShmSlot* slot(int r, uint64_t s) const {
        return reinterpret_cast<ShmSlot*>(base + sizeof(ShmHeader) + ((size_t)r * 2 + (s & 1)) * header()->slot_stride);
    }
};

// Collective over the ranks of one node: rank 0 creates `name` (a POSIX shm name, "/...") and the others attach to it.
// nonce != 0: an attaching rank only accepts a segment whose header carries this value -- a stale segment of a crashed run with the
// same name (magic, world and max_bytes all plausible) is unmapped and the open retried until rank 0 has replaced it.
extern "C" int32_t jolt_shm_create_nonce(const char* name, uint64_t nonce, int32_t rank, int32_t world, size_t max_bytes, jolt_shm** out) {
    if (!name || name[0] != '/' || !out || rank < 0 || world < 1 || rank >= world || max_bytes == 0) return JOLT_ERR_INVALID_ARG;
    jolt_shm* s = new (std::nothrow) jolt_shm();
    if (!s) return JOLT_ERR_OOM;
    s->name = name;
    s->rank = rank;
    s->world = world;
    s->max_bytes = max_bytes;
    const size_t stride = (sizeof(ShmSlot) + max_bytes + 63) & ~(size_t)63;
    s->map_bytes = sizeof(ShmHeader) + (size_t)world * 2 * stride;
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutSeconds; };
    if (rank == 0) {
        (void)shm_unlink(name);  // a stale segment of a crashed run
        int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)s->map_bytes) != 0) { close(fd); fd = -1; (void)shm_unlink(name); }
        if (fd < 0) { delete s; return JOLT_ERR_UNSUPPORTED; }
        s->owner = true;
        void* p = mmap(nullptr, s->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { (void)shm_unlink(name); delete s; return JOLT_ERR_UNSUPPORTED; }
        s->base = static_cast<char*>(p);
        ShmHeader* h = s->header();  // a fresh segment is zero-filled: every seq starts at 0
        h->world = (uint64_t)world;
        h->max_bytes = max_bytes;
        h->slot_stride = stride;
        h->nonce = nonce;
        h->magic.store(kShmMagic, std::memory_order_release);
        *out = s;
        return JOLT_OK;
    }
    for (;;) {  // wait for rank 0 to create, size and initialise THIS run's segment
        int fd = shm_open(name, O_RDWR, 0600);
        if (fd >= 0) {
            struct stat sb;
            void* p = MAP_FAILED;
            if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= s->map_bytes) p = mmap(nullptr, s->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (p != MAP_FAILED) {
                ShmHeader* h = reinterpret_cast<ShmHeader*>(p);
                bool ready = false;
                for (int spin = 0; spin < 2000 && !ready; ++spin) {  // ~ a few ms: rank 0 initialises right after the create
                    ready = h->magic.load(std::memory_order_acquire) == kShmMagic;
                    if (!ready) sched_yield();
                }
                if (ready && (nonce == 0 || h->nonce == nonce)) {
                    if (h->world != (uint64_t)world || h->max_bytes != max_bytes) {
                        munmap(p, s->map_bytes);
                        delete s;
                        return JOLT_ERR_SIZE_MISMATCH;
                    }
                    s->base = static_cast<char*>(p);
                    *out = s;
                    return JOLT_OK;
                }
                munmap(p, s->map_bytes);  // not initialised yet, or another run's segment: look again
            }
        }
        if (timed_out()) { delete s; return JOLT_ERR_UNSUPPORTED; }
        usleep(200);
    }
}

extern "C" int32_t jolt_shm_create(const char* name, int32_t rank, int32_t world, size_t max_bytes, jolt_shm** out) {
    return jolt_shm_create_nonce(name, 0, rank, world, max_bytes, out);
}

extern "C" int32_t jolt_shm_destroy(jolt_shm* s) {
    if (!s) return JOLT_OK;
    if (s->base) munmap(s->base, s->map_bytes);
    if (s->owner && !s->unlinked) (void)shm_unlink(s->name.c_str());
    delete s;
    return JOLT_OK;
}

// gathered = world blocks of `bytes` in rank order; every rank must pass the same `bytes` (<= max_bytes).
extern "C" int32_t jolt_shm_all_gather(jolt_shm* s, const void* local, size_t bytes, void* gathered) {
    if (!s || (!local && bytes) || (!gathered && bytes)) return JOLT_ERR_INVALID_ARG;
    if (bytes > s->max_bytes) return JOLT_ERR_SIZE_MISMATCH;
    const uint64_t seq = ++s->seq;
    ShmSlot* mine = s->slot(s->rank, seq);
    if (bytes) std::memcpy(reinterpret_cast<char*>(mine) + sizeof(ShmSlot), local, bytes);
    mine->bytes = bytes;
    mine->seq.store(seq, std::memory_order_release);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < s->world; ++r) {
        ShmSlot* o = s->slot(r, seq);
        uint64_t spins = 0;
        while (o->seq.load(std::memory_order_acquire) != seq) {
            __builtin_ia32_pause();
            ++spins;
            if ((spins & 0x3FFF) == 0) sched_yield();  // more ranks than free cores: give the one we are waiting for a chance to run
            if ((spins & 0xFFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kShmTimeoutSeconds)
                return JOLT_ERR_HIP;  // a rank stopped taking part: fail instead of spinning forever
        }
        if (o->bytes != bytes) return JOLT_ERR_SIZE_MISMATCH;
        if (bytes) std::memcpy(static_cast<char*>(gathered) + (size_t)r * bytes, reinterpret_cast<const char*>(o) + sizeof(ShmSlot), bytes);
    }
    if (s->owner && !s->unlinked) {  // every rank has taken part in an exchange, i.e. has the segment mapped: the name can go now,
        (void)shm_unlink(s->name.c_str());  // so that a run that is killed later leaves nothing behind in /dev/shm
        s->unlinked = true;
    }
    return JOLT_OK;
}

// A jolt_gather_fn for jolt_host_batch_run with user = jolt_shm*.
extern "C" int32_t jolt_shm_gather_round_sums(void* user, const jolt_fr_t* local, size_t count, jolt_fr_t* gathered) {
    return jolt_shm_all_gather(static_cast<jolt_shm*>(user), local, count * sizeof(jolt_fr_t), gathered);
}
