// AddressSanitizer / UBSan driver for the host library's plane log pass (tests/test_host_asan.py): random streams of
// unchanged / shifted / restarted stacks into a ring small enough to wrap every few steps.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" int drl_planelog_observe(const uint8_t* obs, uint8_t* last, uint8_t* P, int64_t* head, int64_t* q, int E, int HW, int cap, uint8_t* kind_out);
int main() {
    const int E = 3, HW = 84 * 84, cap = 9;
    std::vector<uint8_t> obs(E * HW * 4), last(E * HW * 4, 0), P((size_t)E * cap * HW), kind(E);
    std::vector<int64_t> head(E, 0), q(E, 0);
    srand(1);
    for (auto& b : obs) b = rand();
    long tot = 0;
    for (int t = 0; t < 200; ++t) {
        for (int e = 0; e < E; ++e) {
            int k = rand() % 4;
            uint8_t* o = &obs[(size_t)e * HW * 4];
            if (k < 2) { for (int i = 0; i < HW; ++i) { o[4*i] = o[4*i+1]; o[4*i+1] = o[4*i+2]; o[4*i+2] = o[4*i+3]; o[4*i+3] = rand(); } }
            else if (k == 2) { for (int i = 0; i < HW * 4; ++i) o[i] = rand(); }
        }
        tot += drl_planelog_observe(obs.data(), last.data(), P.data(), head.data(), q.data(), E, HW, cap, kind.data());
    }
    printf("ok %ld head %ld %ld %ld\n", tot, (long)head[0], (long)head[1], (long)head[2]);
    return 0;
}
