// ThreadSanitizer stress harness for the shared-memory runtime ("TSAN build of the C++ shm ring").
// Built by tests/test_host_tsan.py as:  g++ -O1 -g -fsanitize=thread drl_host.cpp stress_tsan.cpp -lrt -pthread
//   * ring: P producer threads push self-describing payloads, ONE consumer checks checksum + per-producer FIFO order
//   * param register: one writer publishes versions of a uniform buffer, readers must never see a torn buffer
//     (the seqlock's racy memcpy is by design and validated by the `writing` word -> suppressed by name, see the test)
//   * heartbeat table: concurrent register / tick / snapshot
// Exit code 0 = all invariants held; TSAN reports go to stderr and fail the test.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <unistd.h>

extern "C" {
void* drl_ring_create(const char* name, uint64_t n_slots, uint64_t slot_bytes);
void drl_ring_close(void* h, int unlink_it);
int drl_ring_push(void* h, const void* data, uint64_t nbytes, uint64_t tag);
void* drl_ring_try_acquire(void* h, uint64_t* ticket_out, uint64_t* nbytes_out, uint64_t* tag_out);
void drl_ring_release(void* h, uint64_t ticket);
uint64_t drl_ring_pushed(void* h);
void* drl_param_create(const char* name, uint64_t capacity);
void drl_param_close(void* h, int unlink_it);
uint64_t drl_param_publish(void* h, const void* data, uint64_t nbytes);
uint64_t drl_param_read(void* h, void* out, uint64_t out_capacity, uint64_t have_version, uint64_t* nbytes_out);
void* drl_beat_open(const char* name, int create);
void drl_beat_close(void* h, const char* name_to_unlink);
int drl_beat_register(void* h, uint64_t pid, uint64_t role, uint64_t ident);
void drl_beat_tick(void* h, int slot, uint64_t progress);
void drl_beat_unregister(void* h, int slot);
int drl_beat_snapshot(void* h, uint64_t* out, int max_rows);
}

static std::atomic<int> failures{0};
#define CHECK(c) do { if (!(c)) { failures.fetch_add(1); fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); } } while (0)

struct Msg { uint32_t producer, seq, len, sum; uint8_t body[240]; };

static uint32_t checksum(const uint8_t* p, uint32_t n) { uint32_t s = 2166136261u; for (uint32_t i = 0; i < n; ++i) s = (s ^ p[i]) * 16777619u; return s; }

static int test_ring(int P, int per) {
    char name[64]; snprintf(name, sizeof name, "/drl_tsan_ring_%d", (int)getpid());
    void* ring = drl_ring_create(name, 64, sizeof(Msg));
    if (!ring) return 1;
    std::vector<std::thread> th;
    for (int p = 0; p < P; ++p)
        th.emplace_back([=] {
            Msg m;
            for (int i = 0; i < per; ++i) {
                m.producer = p; m.seq = i; m.len = 16 + (uint32_t)((p * 31 + i * 7) % 200);
                for (uint32_t k = 0; k < m.len; ++k) m.body[k] = (uint8_t)(p + i + k);
                m.sum = checksum(m.body, m.len);
                while (drl_ring_push(ring, &m, 16 + m.len, (uint64_t)p) != 1) std::this_thread::yield();
            }
        });
    std::vector<uint32_t> next(P, 0);
    long got = 0, want = (long)P * per;
    while (got < want) {
        uint64_t t, nb, tag;
        void* q = drl_ring_try_acquire(ring, &t, &nb, &tag);
        if (!q) { std::this_thread::yield(); continue; }
        Msg m; memcpy(&m, q, nb);
        drl_ring_release(ring, t);
        CHECK(m.producer < (uint32_t)P && tag == m.producer);
        CHECK(nb == 16 + m.len && m.sum == checksum(m.body, m.len));
        if (m.producer < (uint32_t)P) { CHECK(m.seq == next[m.producer]); next[m.producer] = m.seq + 1; }
        ++got;
    }
    for (auto& t : th) t.join();
    CHECK(drl_ring_pushed(ring) == (uint64_t)want);
    drl_ring_close(ring, 1);
    return 0;
}

static int test_param(int versions, int readers) {
    char name[64]; snprintf(name, sizeof name, "/drl_tsan_param_%d", (int)getpid());
    const uint64_t NB = 1 << 14;
    void* par = drl_param_create(name, NB);
    if (!par) return 1;
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int r = 0; r < readers; ++r)
        th.emplace_back([&] {
            std::vector<uint8_t> out(NB);
            uint64_t have = 0, nb = 0;
            while (!stop.load(std::memory_order_acquire) || have < (uint64_t)versions) {
                uint64_t v = drl_param_read(par, out.data(), NB, have, &nb);
                if (v == 0 || v == ~0ull) { std::this_thread::yield(); if (stop.load() && have >= (uint64_t)versions) break; continue; }
                CHECK(v > have && nb == NB);
                const uint8_t b = (uint8_t)(v * 37);
                bool uniform = true;
                for (uint64_t i = 0; i < NB; i += 61) uniform &= out[i] == b;
                CHECK(uniform);              // never a torn / half-overwritten buffer
                have = v;
            }
        });
    std::vector<uint8_t> buf(NB);
    for (int v = 1; v <= versions; ++v) {
        memset(buf.data(), (uint8_t)(v * 37), NB);
        CHECK(drl_param_publish(par, buf.data(), NB) == (uint64_t)v);
        if (v % 8 == 0) std::this_thread::yield();
    }
    stop.store(true, std::memory_order_release);
    for (auto& t : th) t.join();
    drl_param_close(par, 1);
    return 0;
}

static int test_beats(int n) {
    char name[64]; snprintf(name, sizeof name, "/drl_tsan_beat_%d", (int)getpid());
    void* tab = drl_beat_open(name, 1);
    if (!tab) return 1;
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([=] {
            int slot = drl_beat_register(tab, 1000 + i, i % 2, i);
            CHECK(slot >= 0);
            for (uint64_t k = 1; k <= 300; ++k) drl_beat_tick(tab, slot, k);
            drl_beat_unregister(tab, slot);
        });
    uint64_t rows[64 * 8];
    for (int k = 0; k < 50; ++k) { int m = drl_beat_snapshot(tab, rows, 64); CHECK(m >= 0 && m <= 64); }
    for (auto& t : th) t.join();
    drl_beat_close(tab, name);
    return 0;
}

int main(int argc, char** argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 1;           // 1 = quick (CI), 10 = soak
    int rc = test_ring(4, 2000 * scale) | test_param(200 * scale, 3) | test_beats(8);
    if (rc || failures.load()) { fprintf(stderr, "stress FAILED rc=%d failures=%d\n", rc, failures.load()); return 1; }
    printf("stress ok\n");
    return 0;
}
