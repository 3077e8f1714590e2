// tests/cpp/test_exchange.cc -- the all-to-all schedule of superplus_amd/csrc/dfk_exchange.h over host memory, every
// rank a thread (LoopbackTransport): every unit must arrive at the rank it was addressed to, in source-rank order,
// whatever the piece size (several rounds per pair, a different number for every pair) and with empty slices.
//   g++ -O1 -std=c++17 -pthread -o test_exchange tests/cpp/test_exchange.cc && ./test_exchange
#include "../../superplus_amd/csrc/dfk_exchange.h"
#include <cstdio>
#include <random>
#include <thread>

static int run(int world, uint64_t unit, uint64_t piece, unsigned seed)
{
    dfkx::LoopbackHub hub(world);
    std::vector<std::vector<uint64_t>> counts(world, std::vector<uint64_t>(world));
    std::mt19937 rng(seed);
    for (int r = 0; r < world; ++r) for (int d = 0; d < world; ++d) counts[r][d] = rng() % 40;
    if (world > 1) { counts[1][0] = 0; counts[0][world - 1] = 0; }                    // empty slices
    std::vector<int> bad(world, 0);
    std::vector<uint64_t> sums(world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            dfkx::LoopbackTransport T(hub, r);
            // unit (src, dst, serial) so that the receiver can check routing and order
            uint64_t total = 0; for (int d = 0; d < world; ++d) total += counts[r][d];
            std::vector<uint8_t> send(total * unit, 0);
            uint64_t at = 0;
            for (int d = 0; d < world; ++d) for (uint64_t s = 0; s < counts[r][d]; ++s, ++at) { uint8_t* u = &send[at * unit]; u[0] = (uint8_t)r; u[1] = (uint8_t)d; u[2] = (uint8_t)s; for (uint64_t k = 3; k < unit; ++k) u[k] = (uint8_t)(r * 31 + d * 7 + s + k); }
            std::vector<uint64_t> all((size_t)world * world), rc(world);
            T.all_gather(counts[r].data(), world, all.data());
            for (int s = 0; s < world; ++s) rc[s] = all[(size_t)s * world + r];
            uint64_t rtotal = 0; for (int s = 0; s < world; ++s) rtotal += rc[s];
            std::vector<uint8_t> recv(rtotal * unit + 1, 0xEE);
            dfkx::all_to_all_v(T, send.data(), counts[r].data(), recv.data(), rc.data(), unit, piece);
            at = 0;
            for (int s = 0; s < world; ++s) for (uint64_t k = 0; k < rc[s]; ++k, ++at) {
                const uint8_t* u = &recv[at * unit];
                if (u[0] != s || u[1] != r || u[2] != (uint8_t)k) ++bad[r];
                for (uint64_t j = 3; j < unit; ++j) if (u[j] != (uint8_t)(s * 31 + r * 7 + k + j)) { ++bad[r]; break; }
            }
            if (recv[rtotal * unit] != 0xEE) ++bad[r];                               // nothing written past the end
            uint64_t v[2] = {total, (uint64_t)r};
            T.all_reduce(v, 2, false); sums[r] = v[0];
            uint64_t m[1] = {(uint64_t)(r * 10)};
            T.all_reduce(m, 1, true); if (m[0] != (uint64_t)((world - 1) * 10)) ++bad[r];
        });
    for (auto& t : th) t.join();
    uint64_t expect = 0; for (int r = 0; r < world; ++r) for (int d = 0; d < world; ++d) expect += counts[r][d];
    int fails = 0;
    for (int r = 0; r < world; ++r) { fails += bad[r]; if (sums[r] != expect) ++fails; }
    printf("world %d unit %llu piece %llu: %s\n", world, (unsigned long long)unit, (unsigned long long)piece, fails ? "FAILED" : "ok");
    return fails;
}

// all_gather_v: every rank's share arrives at every other rank at its place (sources in rank order, the receiver's own left out)
static int run_gather(int world, uint64_t unit, uint64_t piece, unsigned seed)
{
    dfkx::LoopbackHub hub(world);
    std::mt19937 rng(seed);
    std::vector<uint64_t> counts(world);
    for (int r = 0; r < world; ++r) counts[r] = rng() % 50;
    if (world > 1) counts[world - 1] = 0;                                              // a rank with nothing to give
    std::vector<int> bad(world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            dfkx::LoopbackTransport T(hub, r);
            std::vector<uint8_t> mine(counts[r] * unit + 1, 0);
            for (uint64_t s = 0; s < counts[r]; ++s) for (uint64_t k = 0; k < unit; ++k) mine[s * unit + k] = (uint8_t)(r * 37 + s * 3 + k);
            std::vector<uint64_t> all(world);
            T.all_gather(&counts[r], 1, all.data());
            uint64_t incoming = 0; for (int s = 0; s < world; ++s) { if (all[s] != counts[s]) ++bad[r]; if (s != r) incoming += counts[s]; }
            std::vector<uint8_t> room(incoming * unit + 1, 0xEE);
            dfkx::all_gather_v(T, mine.data(), all.data(), room.data(), unit, piece);
            uint64_t at = 0;
            for (int s = 0; s < world; ++s) {
                if (s == r) continue;
                for (uint64_t k = 0; k < counts[s]; ++k, ++at) for (uint64_t j = 0; j < unit; ++j) if (room[at * unit + j] != (uint8_t)(s * 37 + k * 3 + j)) { ++bad[r]; break; }
            }
            if (room[incoming * unit] != 0xEE) ++bad[r];
        });
    for (auto& t : th) t.join();
    int fails = 0; for (int b : bad) fails += b;
    printf("gather: world %d unit %llu piece %llu: %s\n", world, (unsigned long long)unit, (unsigned long long)piece, fails ? "FAILED" : "ok");
    return fails;
}

int main()
{
    int fails = 0;
    for (int world : {1, 2, 3, 4, 8})
        for (uint64_t piece : {(uint64_t)1 << 30, (uint64_t)64, (uint64_t)32})
            fails += run_gather(world, 32, piece, 300 + world);
    for (int world : {1, 2, 3, 4, 8})
        for (uint64_t piece : {(uint64_t)1 << 30, (uint64_t)96, (uint64_t)40, (uint64_t)32})
            fails += run(world, 32, piece, 100 + world);
    fails += run(4, 16, 48, 7);
    fails += run(5, 4, 12, 9);
    return fails ? 1 : 0;
}
