// drl_host — host-side native runtime of distributed_rl_amd (C ABI, loaded with ctypes).
//
// Replaces the reference's Redis transport for on-node traffic:
//   * ExperienceRing : bounded many-producer / single-consumer queue of fixed-size slots in POSIX
//                      shared memory. Actors write uint8 frame blocks straight into a slot (no pickle,
//                      no lz4, no TCP), the learner's ingest thread consumes them in place and uploads
//                      to HBM. Semantics kept from `RPUSH experience` / `LRANGE+LTRIM` (ape_x/actor.py:97,
//                      ape_x/replay.py:26-29): unordered across producers, at-most-once, atomic drain;
//                      `LLEN` back-pressure (ape_x/learner.py:47-49) becomes ring fill level.
//   * ParamRegister  : latest-value register with a version counter (`SET/GET params`,
//                      ape_x/learner.py:83, ape_x/actor.py:52-56): double-buffered, seqlock-validated, readers may
//                      skip versions and see "nothing yet" (version 0) before the first publish.
//   * HeartbeatTable : per-process liveness counters for the launcher's watchdog.
//
// All cross-process state is plain std::atomic<uint64_t> in the mapping (lock-free on x86-64), using the
// per-slot sequence protocol of a bounded MPMC queue: slot.seq == ticket        -> free for producer `ticket`
//                                                    slot.seq == ticket + 1    -> filled, consumer may read
//                                                    slot.seq == ticket + N    -> released for the next lap.
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <new>
#include <csignal>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr uint64_t kRingMagic = 0x44524c52494e4731ull;   // "DRLRING1"
constexpr uint64_t kParamMagic = 0x44524c5041524d31ull;  // "DRLPARM1"
constexpr uint64_t kBeatMagic = 0x44524c4245415431ull;   // "DRLBEAT1"
constexpr size_t kCacheLine = 64;

inline size_t align_up(size_t n, size_t a) { return (n + a - 1) / a * a; }

struct Mapping {
    void* base = nullptr;
    size_t bytes = 0;
    int fd = -1;
};

bool map_shm(const char* name, size_t bytes, bool create, Mapping* out) {
    int flags = create ? (O_CREAT | O_RDWR | O_EXCL) : O_RDWR;
    int fd = shm_open(name, flags, 0600);
    if (fd < 0 && create && errno == EEXIST) {  // stale segment from a dead run: replace it
        shm_unlink(name);
        fd = shm_open(name, flags, 0600);
    }
    if (fd < 0) return false;
    if (create) {
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return false; }
    } else {
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return false; }
        bytes = (size_t)st.st_size;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) { close(fd); if (create) shm_unlink(name); return false; }
    out->base = p; out->bytes = bytes; out->fd = fd;
    return true;
}

void unmap(Mapping* m) {
    if (m->base) munmap(m->base, m->bytes);
    if (m->fd >= 0) close(m->fd);
    m->base = nullptr; m->fd = -1;
}

uint64_t now_ns() {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// ------------------------------------------------------------------------------------------------
struct alignas(kCacheLine) RingHeader {
    uint64_t magic;
    uint64_t n_slots;       // power of two
    uint64_t slot_bytes;    // payload capacity per slot
    uint64_t slot_stride;   // header + payload, cache-line aligned
    alignas(kCacheLine) std::atomic<uint64_t> tail;      // next producer ticket
    alignas(kCacheLine) std::atomic<uint64_t> head;      // next consumer ticket (single consumer)
    alignas(kCacheLine) std::atomic<uint64_t> pushed;    // committed blocks (stat)
    std::atomic<uint64_t> rejected;                      // try_reserve failures because full (stat)
    std::atomic<uint64_t> epoch;                         // bumped by ring_reset (DEL experience)
    std::atomic<uint64_t> abandoned;                     // reservations of dead producers skipped by the consumer (stat)
};

struct alignas(kCacheLine) SlotHeader {
    std::atomic<uint64_t> seq;
    uint64_t nbytes;
    uint64_t tag;           // user tag (block kind / producer id)
    // who holds the reservation: written by the producer right after it has claimed the ticket, read by the consumer
    // only when the head slot has stayed reserved-but-uncommitted for longer than the grace period (see try_acquire)
    std::atomic<uint64_t> owner_pid;
    std::atomic<uint64_t> owner_ticket;    // ticket + 1 (0 = never reserved)
};
static_assert(sizeof(SlotHeader) == kCacheLine, "slot header is one cache line");

struct Ring {
    Mapping map;
    RingHeader* hdr;
    uint8_t* slots;
    std::string name;
    bool owner;
    uint64_t stall_pos = ~0ull, stall_since_ns = 0;      // consumer-local: head ticket seen reserved-but-uncommitted since
    uint64_t abandon_grace_ns = 2000000000ull;           // how long before the reservation's owner is checked for life
    SlotHeader* slot(uint64_t ticket) const {
        return reinterpret_cast<SlotHeader*>(slots + (ticket & (hdr->n_slots - 1)) * hdr->slot_stride);
    }
};

// ------------------------------------------------------------------------------------------------
struct alignas(kCacheLine) ParamHeader {
    uint64_t magic;
    uint64_t capacity;                                   // bytes per buffer
    alignas(kCacheLine) std::atomic<uint64_t> version;   // 0 = nothing published; buffer = version & 1
    std::atomic<uint64_t> nbytes[2];
    std::atomic<uint64_t> writing;                       // version currently being written (seqlock guard)
};

// Device mirror of the parameter register (second kilobyte of the header page): the learner keeps the two newest
// versions in a double buffer in ITS GPU's memory and exports it by IPC handle; actors on the same GPU copy device to
// device instead of D2H -> shm -> H2D. slot_ver[v & 1] == v says that slot holds version v completely (0: being written).
struct alignas(kCacheLine) MirrorRecord {
    std::atomic<uint64_t> valid;                         // 0 = no mirror; else the exporting process' pid
    uint64_t slot_bytes;
    uint8_t handle[64];                                  // hipIpcMemHandle_t of the double buffer
    char bus_id[64];                                     // PCI bus id of the GPU that holds it
    alignas(kCacheLine) std::atomic<uint64_t> latest;
    std::atomic<uint64_t> slot_ver[2];
};
constexpr size_t kMirrorOffset = 1024;
static_assert(sizeof(ParamHeader) <= kMirrorOffset && kMirrorOffset + sizeof(MirrorRecord) <= 4096, "header page layout");

struct Param {
    Mapping map;
    ParamHeader* hdr;
    uint8_t* buf[2];
    std::string name;
};
static MirrorRecord* mirror_of(void* h) {
    return reinterpret_cast<MirrorRecord*>(static_cast<uint8_t*>(static_cast<Param*>(h)->map.base) + kMirrorOffset);
}

// ------------------------------------------------------------------------------------------------
constexpr int kMaxBeats = 1024;
struct alignas(kCacheLine) BeatSlot {
    std::atomic<uint64_t> pid;
    std::atomic<uint64_t> role;      // 0 free, 1 learner, 2 actor, 3 replay, 4 other
    std::atomic<uint64_t> ident;     // actor number etc.
    std::atomic<uint64_t> beats;
    std::atomic<uint64_t> last_ns;
    std::atomic<uint64_t> progress;  // user counter (env steps / learner steps)
};
struct BeatHeader { uint64_t magic; uint64_t n; };
struct Beat { Mapping map; BeatHeader* hdr; BeatSlot* slots; };

}  // namespace

extern "C" {

// ---- ExperienceRing ---------------------------------------------------------------------------------
void* drl_ring_create(const char* name, uint64_t n_slots, uint64_t slot_bytes) {
    uint64_t n = 1; while (n < n_slots) n <<= 1;
    size_t stride = align_up(sizeof(SlotHeader) + slot_bytes, kCacheLine);
    size_t total = align_up(sizeof(RingHeader), 4096) + stride * n;
    Ring* r = new (std::nothrow) Ring();
    if (!r) return nullptr;
    if (!map_shm(name, total, true, &r->map)) { delete r; return nullptr; }
    r->hdr = new (r->map.base) RingHeader();
    r->hdr->n_slots = n; r->hdr->slot_bytes = slot_bytes; r->hdr->slot_stride = stride;
    r->hdr->tail.store(0); r->hdr->head.store(0); r->hdr->pushed.store(0); r->hdr->rejected.store(0);
    r->hdr->epoch.store(1); r->hdr->abandoned.store(0);
    r->slots = static_cast<uint8_t*>(r->map.base) + align_up(sizeof(RingHeader), 4096);
    for (uint64_t i = 0; i < n; ++i) {
        SlotHeader* s = new (r->slots + i * stride) SlotHeader();
        s->seq.store(i, std::memory_order_relaxed); s->nbytes = 0; s->tag = 0;
        s->owner_pid.store(0, std::memory_order_relaxed); s->owner_ticket.store(0, std::memory_order_relaxed);
    }
    std::atomic_thread_fence(std::memory_order_release);
    r->hdr->magic = kRingMagic;
    r->name = name; r->owner = true;
    return r;
}

void* drl_ring_open(const char* name) {
    Ring* r = new (std::nothrow) Ring();
    if (!r) return nullptr;
    if (!map_shm(name, 0, false, &r->map)) { delete r; return nullptr; }
    r->hdr = static_cast<RingHeader*>(r->map.base);
    if (r->map.bytes < sizeof(RingHeader) || r->hdr->magic != kRingMagic) { unmap(&r->map); delete r; return nullptr; }
    r->slots = static_cast<uint8_t*>(r->map.base) + align_up(sizeof(RingHeader), 4096);
    r->name = name; r->owner = false;
    return r;
}

void drl_ring_close(void* h, int unlink_it) {
    Ring* r = static_cast<Ring*>(h);
    if (!r) return;
    unmap(&r->map);
    if (unlink_it) shm_unlink(r->name.c_str());
    delete r;
}

uint64_t drl_ring_slot_bytes(void* h) { return static_cast<Ring*>(h)->hdr->slot_bytes; }
uint64_t drl_ring_n_slots(void* h) { return static_cast<Ring*>(h)->hdr->n_slots; }

// Producer step 1: claim a free slot. Returns payload pointer or NULL when the ring is full.
void* drl_ring_try_reserve(void* h, uint64_t* ticket_out) {
    Ring* r = static_cast<Ring*>(h);
    uint64_t pos = r->hdr->tail.load(std::memory_order_relaxed);
    for (;;) {
        SlotHeader* s = r->slot(pos);
        uint64_t seq = s->seq.load(std::memory_order_acquire);
        int64_t dif = (int64_t)seq - (int64_t)pos;
        if (dif == 0) {
            if (r->hdr->tail.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
                s->owner_pid.store((uint64_t)getpid(), std::memory_order_relaxed);
                s->owner_ticket.store(pos + 1, std::memory_order_release);
                *ticket_out = pos;
                return reinterpret_cast<uint8_t*>(s) + sizeof(SlotHeader);
            }
        } else if (dif < 0) {
            r->hdr->rejected.fetch_add(1, std::memory_order_relaxed);
            return nullptr;  // full
        } else {
            pos = r->hdr->tail.load(std::memory_order_relaxed);
        }
    }
}

// Producer step 2: publish the filled slot. Returns 1, or 0 when the consumer gave this reservation up meanwhile (a producer
// that stalled for longer than the grace period between its two steps and could not be found alive: the slot has been
// recycled, the block is lost and the producer simply pushes it again). The compare-exchange keeps such a late commit from
// overwriting the sequence number of a slot that already belongs to a later lap.
int drl_ring_commit(void* h, uint64_t ticket, uint64_t nbytes, uint64_t tag) {
    Ring* r = static_cast<Ring*>(h);
    SlotHeader* s = r->slot(ticket);
    s->nbytes = nbytes; s->tag = tag;
    uint64_t expect = ticket;
    if (!s->seq.compare_exchange_strong(expect, ticket + 1, std::memory_order_release, std::memory_order_relaxed)) return 0;
    r->hdr->pushed.fetch_add(1, std::memory_order_relaxed);
    return 1;
}

// Convenience: reserve + memcpy + commit. Returns 1 on success, 0 if full, -1 if too large.
int drl_ring_push(void* h, const void* data, uint64_t nbytes, uint64_t tag) {
    Ring* r = static_cast<Ring*>(h);
    if (nbytes > r->hdr->slot_bytes) return -1;
    uint64_t t;
    void* p = drl_ring_try_reserve(h, &t);
    if (!p) return 0;
    memcpy(p, data, nbytes);
    return drl_ring_commit(h, t, nbytes, tag);          // 0: the reservation was given up meanwhile, push again
}

// Consumer step 1 (single consumer): peek the oldest committed slot. NULL when empty (or when the
// producer holding the head ticket has reserved but not yet committed).
//
// A producer that dies between reserve and commit would wedge the queue for good (the head slot never becomes
// readable, and restarted actors only ever see "full"). The consumer therefore watches the head ticket: when it has
// been claimed (tail > head) but uncommitted for longer than the grace period, the reservation's owner is looked up -
// a live producer wrote its pid and ticket into the slot header microseconds after claiming it - and if that process
// no longer exists (or died before it could even sign the slot), the slot is skipped as an empty block.
void* drl_ring_try_acquire(void* h, uint64_t* ticket_out, uint64_t* nbytes_out, uint64_t* tag_out) {
    Ring* r = static_cast<Ring*>(h);
    for (;;) {
        uint64_t pos = r->hdr->head.load(std::memory_order_relaxed);
        SlotHeader* s = r->slot(pos);
        uint64_t seq = s->seq.load(std::memory_order_acquire);
        if ((int64_t)seq - (int64_t)(pos + 1) >= 0) {
            r->stall_pos = ~0ull;
            *ticket_out = pos; *nbytes_out = s->nbytes; *tag_out = s->tag;
            return reinterpret_cast<uint8_t*>(s) + sizeof(SlotHeader);
        }
        if (r->hdr->tail.load(std::memory_order_acquire) <= pos) { r->stall_pos = ~0ull; return nullptr; }   // empty
        const uint64_t now = now_ns();
        if (r->stall_pos != pos) { r->stall_pos = pos; r->stall_since_ns = now; return nullptr; }
        if (now - r->stall_since_ns < r->abandon_grace_ns) return nullptr;
        const uint64_t ot = s->owner_ticket.load(std::memory_order_acquire);
        const uint64_t pid = s->owner_pid.load(std::memory_order_relaxed);
        const bool signed_by_claimant = ot == pos + 1;
        const bool alive = signed_by_claimant && pid != 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM);
        if (alive) { r->stall_since_ns = now; return nullptr; }      // slow, not dead: look again after another grace period
        // claim the skip with a compare-exchange on the sequence number: a commit that lands at this very moment wins
        // (the loop then finds the slot readable), a commit that comes later fails cleanly (drl_ring_commit)
        uint64_t expect = pos;
        if (!s->seq.compare_exchange_strong(expect, pos + r->hdr->n_slots, std::memory_order_acq_rel, std::memory_order_acquire))
            continue;
        r->hdr->abandoned.fetch_add(1, std::memory_order_relaxed);
        r->hdr->head.store(pos + 1, std::memory_order_relaxed);
        r->stall_pos = ~0ull;
    }
}

// Consumer step 2: hand the slot back to producers.
void drl_ring_release(void* h, uint64_t ticket) {
    Ring* r = static_cast<Ring*>(h);
    SlotHeader* s = r->slot(ticket);
    r->hdr->head.store(ticket + 1, std::memory_order_relaxed);
    s->seq.store(ticket + r->hdr->n_slots, std::memory_order_release);
}

// Approximate fill level (LLEN).
uint64_t drl_ring_size(void* h) {
    Ring* r = static_cast<Ring*>(h);
    uint64_t t = r->hdr->tail.load(std::memory_order_relaxed), hd = r->hdr->head.load(std::memory_order_relaxed);
    return t >= hd ? t - hd : 0;
}
uint64_t drl_ring_pushed(void* h) { return static_cast<Ring*>(h)->hdr->pushed.load(std::memory_order_relaxed); }
uint64_t drl_ring_rejected(void* h) { return static_cast<Ring*>(h)->hdr->rejected.load(std::memory_order_relaxed); }
uint64_t drl_ring_abandoned(void* h) { return static_cast<Ring*>(h)->hdr->abandoned.load(std::memory_order_relaxed); }
// consumer-side handle only: grace period before a stuck reservation's owner is checked (tests shorten it)
void drl_ring_set_abandon_grace(void* h, double seconds) { static_cast<Ring*>(h)->abandon_grace_ns = (uint64_t)(seconds * 1e9); }

// Drop everything queued (consumer side; DEL experience at learner start-up, ape_x/replay.py:21).
uint64_t drl_ring_drain(void* h) {
    uint64_t n = 0, t, nb, tag;
    while (drl_ring_try_acquire(h, &t, &nb, &tag)) { drl_ring_release(h, t); ++n; }
    static_cast<Ring*>(h)->hdr->epoch.fetch_add(1, std::memory_order_relaxed);
    return n;
}

// ---- ParamRegister ----------------------------------------------------------------------------------
void* drl_param_create(const char* name, uint64_t capacity) {
    size_t cap = align_up(capacity, 4096);
    size_t total = 4096 + 2 * cap;
    Param* p = new (std::nothrow) Param();
    if (!p) return nullptr;
    if (!map_shm(name, total, true, &p->map)) { delete p; return nullptr; }
    p->hdr = new (p->map.base) ParamHeader();
    p->hdr->capacity = cap; p->hdr->version.store(0); p->hdr->nbytes[0].store(0); p->hdr->nbytes[1].store(0);
    p->hdr->writing.store(0);
    std::atomic_thread_fence(std::memory_order_release);
    p->hdr->magic = kParamMagic;
    p->buf[0] = static_cast<uint8_t*>(p->map.base) + 4096; p->buf[1] = p->buf[0] + cap;
    p->name = name;
    return p;
}

void* drl_param_open(const char* name) {
    Param* p = new (std::nothrow) Param();
    if (!p) return nullptr;
    if (!map_shm(name, 0, false, &p->map)) { delete p; return nullptr; }
    p->hdr = static_cast<ParamHeader*>(p->map.base);
    if (p->map.bytes < 4096 || p->hdr->magic != kParamMagic) { unmap(&p->map); delete p; return nullptr; }
    p->buf[0] = static_cast<uint8_t*>(p->map.base) + 4096; p->buf[1] = p->buf[0] + p->hdr->capacity;
    p->name = name;
    return p;
}

void drl_param_close(void* h, int unlink_it) {
    Param* p = static_cast<Param*>(h);
    if (!p) return;
    unmap(&p->map);
    if (unlink_it) shm_unlink(p->name.c_str());
    delete p;
}

uint64_t drl_param_capacity(void* h) { return static_cast<Param*>(h)->hdr->capacity; }
uint64_t drl_param_version(void* h) { return static_cast<Param*>(h)->hdr->version.load(std::memory_order_acquire); }
// payload size of version v (valid while v is one of the two newest versions)
uint64_t drl_param_nbytes(void* h, uint64_t v) { return static_cast<Param*>(h)->hdr->nbytes[v & 1].load(std::memory_order_acquire); }

// Single writer. Returns the new version, or 0 if too large.
uint64_t drl_param_publish(void* h, const void* data, uint64_t nbytes) {
    Param* p = static_cast<Param*>(h);
    if (nbytes > p->hdr->capacity) return 0;
    uint64_t v = p->hdr->version.load(std::memory_order_relaxed) + 1;
    p->hdr->writing.store(v, std::memory_order_release);   // readers of buffer v&1 (version v-2) must retry
    std::atomic_thread_fence(std::memory_order_seq_cst);
    memcpy(p->buf[v & 1], data, nbytes);
    p->hdr->nbytes[v & 1].store(nbytes, std::memory_order_relaxed);
    p->hdr->version.store(v, std::memory_order_release);
    return v;
}

// Reset to "nothing published" (DEL params, ape_x/learner.py:37).
void drl_param_reset(void* h) {
    Param* p = static_cast<Param*>(h);
    p->hdr->writing.store(0, std::memory_order_relaxed);
    p->hdr->version.store(0, std::memory_order_release);
    MirrorRecord* m = mirror_of(h);
    m->latest.store(0, std::memory_order_relaxed);
    m->slot_ver[0].store(0, std::memory_order_relaxed); m->slot_ver[1].store(0, std::memory_order_release);
}

// Copy the latest version into `out` if it is newer than `have_version`.
// Returns the version read (>have_version), 0 if nothing newer, or (uint64)-1 if `out` is too small.
uint64_t drl_param_read(void* h, void* out, uint64_t out_capacity, uint64_t have_version, uint64_t* nbytes_out) {
    Param* p = static_cast<Param*>(h);
    for (int attempt = 0; attempt < 64; ++attempt) {
        uint64_t v = p->hdr->version.load(std::memory_order_acquire);
        if (v == 0 || v <= have_version) return 0;
        uint64_t nb = p->hdr->nbytes[v & 1].load(std::memory_order_relaxed);
        if (nb > out_capacity) return ~0ull;
        memcpy(out, p->buf[v & 1], nb);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        uint64_t w = p->hdr->writing.load(std::memory_order_acquire);
        if (w < v + 2) { *nbytes_out = nb; return v; }   // the writer has not started overwriting our buffer
    }
    return 0;
}

// ---- device mirror of the register (see MirrorRecord) -------------------------------------------------
void drl_param_mirror_set(void* h, const void* handle64, const char* bus_id, uint64_t slot_bytes, uint64_t pid) {
    MirrorRecord* m = mirror_of(h);
    m->valid.store(0, std::memory_order_release);
    m->latest.store(0, std::memory_order_relaxed);
    m->slot_ver[0].store(0, std::memory_order_relaxed); m->slot_ver[1].store(0, std::memory_order_relaxed);
    memcpy(m->handle, handle64, 64);
    memset(m->bus_id, 0, sizeof(m->bus_id));
    strncpy(m->bus_id, bus_id, sizeof(m->bus_id) - 1);
    m->slot_bytes = slot_bytes;
    m->valid.store(pid ? pid : 1, std::memory_order_release);
}
void drl_param_mirror_clear(void* h) { mirror_of(h)->valid.store(0, std::memory_order_release); }
// returns the exporter's pid (0: none)
uint64_t drl_param_mirror_get(void* h, void* handle64_out, char* bus_id_out64, uint64_t* slot_bytes_out) {
    MirrorRecord* m = mirror_of(h);
    uint64_t pid = m->valid.load(std::memory_order_acquire);
    if (!pid) return 0;
    memcpy(handle64_out, m->handle, 64);
    memcpy(bus_id_out64, m->bus_id, 64);
    *slot_bytes_out = m->slot_bytes;
    return m->valid.load(std::memory_order_acquire) == pid ? pid : 0;
}
// writer: begin(v) before the device copy into slot v & 1 is enqueued, end(v) after it has COMPLETED
void drl_param_mirror_begin(void* h, uint64_t v) {
    mirror_of(h)->slot_ver[v & 1].store(0, std::memory_order_seq_cst);
}
void drl_param_mirror_end(void* h, uint64_t v) {
    MirrorRecord* m = mirror_of(h);
    m->slot_ver[v & 1].store(v, std::memory_order_release);
    m->latest.store(v, std::memory_order_release);
}
uint64_t drl_param_mirror_latest(void* h) { return mirror_of(h)->latest.load(std::memory_order_acquire); }
// reader, after its copy out of slot v & 1 has completed: 1 if the slot held version v the whole time
int drl_param_mirror_check(void* h, uint64_t v) {
    std::atomic_thread_fence(std::memory_order_seq_cst);
    return mirror_of(h)->slot_ver[v & 1].load(std::memory_order_acquire) == v ? 1 : 0;
}

// ---- HeartbeatTable ---------------------------------------------------------------------------------
void* drl_beat_open(const char* name, int create) {
    size_t total = 4096 + sizeof(BeatSlot) * kMaxBeats;
    Beat* b = new (std::nothrow) Beat();
    if (!b) return nullptr;
    if (!map_shm(name, total, create != 0, &b->map)) { delete b; return nullptr; }
    b->hdr = static_cast<BeatHeader*>(b->map.base);
    b->slots = reinterpret_cast<BeatSlot*>(static_cast<uint8_t*>(b->map.base) + 4096);
    if (create) {
        memset(b->map.base, 0, total);
        b->hdr->n = kMaxBeats;
        std::atomic_thread_fence(std::memory_order_release);
        b->hdr->magic = kBeatMagic;
    } else if (b->hdr->magic != kBeatMagic) { unmap(&b->map); delete b; return nullptr; }
    return b;
}

void drl_beat_close(void* h, const char* name_to_unlink) {
    Beat* b = static_cast<Beat*>(h);
    if (!b) return;
    unmap(&b->map);
    if (name_to_unlink && name_to_unlink[0]) shm_unlink(name_to_unlink);
    delete b;
}

// role 0 = free; kBeatClaiming = a process is filling the slot in: pid / ident / time are written BEFORE the role is
// published, so a snapshot never pairs a live role with the previous owner's pid or a zero pid.
constexpr uint64_t kBeatClaiming = ~0ull;
int drl_beat_register(void* h, uint64_t pid, uint64_t role, uint64_t ident) {
    Beat* b = static_cast<Beat*>(h);
    for (int i = 0; i < kMaxBeats; ++i) {
        uint64_t expect = 0;
        if (b->slots[i].role.compare_exchange_strong(expect, kBeatClaiming)) {
            b->slots[i].pid.store(pid); b->slots[i].ident.store(ident);
            b->slots[i].beats.store(0); b->slots[i].progress.store(0); b->slots[i].last_ns.store(now_ns());
            b->slots[i].role.store(role, std::memory_order_release);
            return i;
        }
    }
    return -1;
}

// Supervisor side: free every slot registered by `pid` (a child that was killed never unregisters). Returns the count.
int drl_beat_free_pid(void* h, uint64_t pid) {
    Beat* b = static_cast<Beat*>(h);
    int n = 0;
    for (int i = 0; i < kMaxBeats; ++i) {
        const uint64_t role = b->slots[i].role.load(std::memory_order_acquire);
        if (role != 0 && role != kBeatClaiming && b->slots[i].pid.load() == pid) {
            b->slots[i].pid.store(0);
            b->slots[i].role.store(0, std::memory_order_release);
            ++n;
        }
    }
    return n;
}

void drl_beat_tick(void* h, int slot, uint64_t progress) {
    Beat* b = static_cast<Beat*>(h);
    if (slot < 0 || slot >= kMaxBeats) return;
    b->slots[slot].progress.store(progress, std::memory_order_relaxed);
    b->slots[slot].beats.fetch_add(1, std::memory_order_relaxed);
    b->slots[slot].last_ns.store(now_ns(), std::memory_order_release);
}

void drl_beat_unregister(void* h, int slot) {
    Beat* b = static_cast<Beat*>(h);
    if (slot < 0 || slot >= kMaxBeats) return;
    b->slots[slot].role.store(0, std::memory_order_release);
}

// out: 6 uint64 per slot {pid, role, ident, beats, age_ns, progress}; returns number of live slots written.
int drl_beat_snapshot(void* h, uint64_t* out, int max_rows) {
    Beat* b = static_cast<Beat*>(h);
    int n = 0;
    uint64_t now = now_ns();
    for (int i = 0; i < kMaxBeats && n < max_rows; ++i) {
        uint64_t role = b->slots[i].role.load(std::memory_order_acquire);
        if (!role || role == kBeatClaiming) continue;
        uint64_t last = b->slots[i].last_ns.load(std::memory_order_acquire);
        out[6 * n + 0] = b->slots[i].pid.load(); out[6 * n + 1] = role; out[6 * n + 2] = b->slots[i].ident.load();
        out[6 * n + 3] = b->slots[i].beats.load(); out[6 * n + 4] = now > last ? now - last : 0;
        out[6 * n + 5] = b->slots[i].progress.load();
        ++n;
    }
    return n;
}

int drl_shm_unlink(const char* name) { return shm_unlink(name); }

// ---- PlaneLog.observe (algo/vec_assembly.py) for stacks of 4 uint8 planes, all environments of an actor in one call ----
// Per environment: is the observation the one seen last (nothing happens), that one moved on by ONE frame (its newest
// plane enters the env's plane ring), or unrelated (all four planes enter)? Decided on the pixels in one pass over the two
// stacks (numpy: a compare pass + three passes over a scratch array, a strided gather of the new plane and a copy of the
// stack: 0.5 ms of a 1.3 ms actor tick at 64 envs). obs / last [E][HW][4] u8 (HWC), P [E][cap][HW], head / q [E] int64
// (planes appended so far / stream position of the current observation), kind_out [E]: 0 same, 1 shifted, 2 restarted.
// (no clones under the thread sanitizer: their ifunc resolvers run before its runtime is up)
#if defined(__GNUC__) && !defined(__clang__) && defined(__x86_64__) && !defined(__SANITIZE_THREAD__)
#define DRL_VEC __attribute__((target_clones("avx2", "default"), optimize("O3")))
#else
#define DRL_VEC
#endif
DRL_VEC static void plane_relation(const uint32_t* a, const uint32_t* b, int n, uint32_t* d_same, uint32_t* d_shift) {
    uint32_t ds = 0, dh = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t x = a[i], y = b[i];
        ds |= x ^ y;
        dh |= ((y >> 8) ^ x) & 0x00FFFFFFu;                    // planes 0..2 of obs == planes 1..3 of the last one
    }
    *d_same = ds; *d_shift = dh;
}
DRL_VEC static void plane_extract(const uint32_t* a, int n, int c, uint8_t* dst) {
    const int sh = 8 * c;
    for (int i = 0; i < n; ++i) dst[i] = (uint8_t)(a[i] >> sh);
}
int drl_planelog_observe(const uint8_t* obs, uint8_t* last, uint8_t* P, int64_t* head, int64_t* q, int E, int HW, int cap,
                         uint8_t* kind_out) {
    if (!obs || !last || !P || !head || !q || E <= 0 || HW <= 0 || cap < 4) return -1;
    int changed = 0;
    for (int e = 0; e < E; ++e) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(obs + (size_t)e * HW * 4);
        uint32_t* b = reinterpret_cast<uint32_t*>(last + (size_t)e * HW * 4);
        uint32_t ds, dh;
        plane_relation(a, b, HW, &ds, &dh);
        uint8_t kind = 0;
        if (ds != 0) {
            uint8_t* ring = P + (size_t)e * cap * HW;
            if (dh == 0) {                                      // moved on by one frame: one new plane
                plane_extract(a, HW, 3, ring + (size_t)(head[e] % cap) * HW);
                head[e] += 1; q[e] += 1;
                kind = 1;
            } else {                                            // unrelated to what the env showed before: all four planes
                q[e] = head[e];
                for (int c = 0; c < 4; ++c) {
                    plane_extract(a, HW, c, ring + (size_t)(head[e] % cap) * HW);
                    head[e] += 1;
                }
                kind = 2;
            }
            memcpy(b, a, (size_t)HW * 4);
            ++changed;
        }
        if (kind_out) kind_out[e] = kind;
    }
    return changed;
}

const char* drl_host_version() { return "drl_host 0.3"; }

}  // extern "C"
