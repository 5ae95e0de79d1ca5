"""CPU suite: the index arithmetic of the prefill-GEMM candidate (jukebox_amd/csrc/gemm_glds_index.h, used by gemm_glds_kernel in gemm.hip and by tools/gemm_glds_probe.hip)
checked exhaustively with a g++-compiled harness -- no GPU involved:
  * what the LDS-DMA instructions deposit (lane-linear) is what the operand reads expect, every (row, segment) exactly once;
  * every ds_read_b128 of an MFMA operand is bank-conflict free for the lane groups MI355X services together
    (MI355X_MICROARCH.md, LDS: 4 groups of 16 lanes, 64 banks of 4 bytes);
  * every LDS-DMA instruction of the activation panel fetches whole 128-byte lines;
  * the XCD-aware tile order visits every tile exactly once and keeps a row panel on one XCD."""
import os
import subprocess

from conftest import ROOT

HARNESS = r"""
#include <cstdio>
#include <set>
#include <vector>
#include <map>
#include "gemm_glds_index.h"
using namespace gi;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main() {
    // 1. deposit == read formula, full coverage
    std::set<std::pair<int, int>> seen;
    for (int piece = 0; piece < A_PIECES; ++piece)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = piece * 8 + a_src_row(lane), seg = a_src_seg(lane);
            CHECK(seg >= 0 && seg < 8);
            CHECK(a_byte(row, seg) == piece * 1024 + lane * 16);
            CHECK(seen.insert({row, seg}).second);
        }
    CHECK((int)seen.size() == BM * 8);
    // whole lines per instruction: the 8 lanes of a row fetch segments 0..7
    for (int r = 0; r < 8; ++r) {
        int mask = 0;
        for (int lane = r * 8; lane < r * 8 + 8; ++lane) { CHECK(a_src_row(lane) == r); mask |= 1 << a_src_seg(lane); }
        CHECK(mask == 255);
    }
    // 2. bank conflicts of the operand reads (ds_read_b128: 4 lane groups, each must touch the 64 banks once)
    const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                               {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                               {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                               {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int wave_m = 0; wave_m < 2; ++wave_m)
        for (int mt = 0; mt < 4; ++mt)
            for (int ks = 0; ks < 2; ++ks)
                for (int g = 0; g < 4; ++g) {
                    std::set<int> banks;
                    for (int i = 0; i < 16; ++i) {
                        const int lane = groups[g][i];
                        const int a = a_byte(frag_row(wave_m, mt, lane), frag_seg(ks, lane));
                        CHECK(a % 16 == 0 && a >= 0 && a + 16 <= A_BYTES);
                        for (int w = 0; w < 4; ++w) CHECK(banks.insert((a / 4 + w) % 64).second);
                    }
                }
    for (int jt = 0; jt < BJT; ++jt)
        for (int ks = 0; ks < 2; ++ks) {
            CHECK(w_byte(jt, ks, 0) == A_BYTES + (jt * 2 + ks) * 1024 && w_byte(jt, ks, 63) + 16 <= STAGE_BYTES);
            for (int g = 0; g < 4; ++g) {
                std::set<int> banks;
                for (int i = 0; i < 16; ++i)
                    for (int w = 0; w < 4; ++w) CHECK(banks.insert((w_byte(jt, ks, groups[g][i]) / 4 + w) % 64).second);
            }
        }
    // 3. tile order
    const int shapes[][2] = {{256, 15}, {256, 12}, {250, 23}, {3, 5}, {8, 1}, {1, 1}, {17, 38}};
    for (auto& s : shapes) {
        const int MB = s[0], NB = s[1], grid = (MB + 7) / 8 * 8 * NB;
        std::set<std::pair<int, int>> tiles;
        std::map<int, int> xcd_of_panel;
        for (int b = 0; b < grid; ++b) {
            int mp, nt;
            if (!tile_of_block(b, MB, NB, &mp, &nt)) continue;
            CHECK(mp >= 0 && nt >= 0 && nt < NB);
            CHECK(tiles.insert({mp, nt}).second);
            auto it = xcd_of_panel.find(mp);
            if (it == xcd_of_panel.end()) xcd_of_panel[mp] = b & 7; else CHECK(it->second == (b & 7));
        }
        CHECK((int)tiles.size() == MB * NB);
    }
    std::printf("ok\n");
    return 0;
}
"""


def test_gemm_glds_index_arithmetic(tmp_path):
    src = tmp_path / "harness.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "harness"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "jukebox_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


HARNESS_8PHASE = r"""
#include <cstdio>
#include <set>
#include <map>
#include <vector>
#include "gemm_8phase.h"
using namespace g8;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                  {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                  {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                  {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
int main() {
    // 1. the eight half-tiles of the two stages tile the 128 KiB without overlap
    std::set<int> bases;
    for (int st = 0; st < 2; ++st)
        for (int kind = 0; kind < 2; ++kind)
            for (int h = 0; h < 2; ++h) {
                const int b = half_base(st, kind, h);
                CHECK(b % HALF_BYTES == 0 && b >= 0 && b + HALF_BYTES <= LDS_BYTES && bases.insert(b).second);
            }
    // 2. activations: what the 8 waves' LDS-DMA requests deposit (wave w: pieces 2w, 2w + 1 of a half, lane-linear) is what the
    //    operand reads expect, and the two halves hold every (tile row, segment) exactly once
    std::set<std::pair<int, int>> seen;
    for (int h = 0; h < 2; ++h)
        for (int wave = 0; wave < 8; ++wave)
            for (int u = 0; u < 2; ++u)
                for (int lane = 0; lane < 64; ++lane) {
                    const int rl = (wave * 2 + u) * 8 + gi::a_src_row(lane), seg = gi::a_src_seg(lane);
                    CHECK(a_half_byte(rl, seg) == wave * 2048 + u * 1024 + lane * 16);          // dst = half + wave * 2048 (+ 1024)
                    const int row = a_tile_row(h, rl);
                    CHECK(row >= 0 && row < BM && seen.insert({row, seg}).second);
                    // the wave row that reads it: rows [128 wm, 128 wm + 128), quadrant row h = rows 64 h .. 64 h + 63 of those
                    CHECK((row >> 7) == (rl >> 6) && ((row >> 6) & 1) == h);
                }
    CHECK((int)seen.size() == BM * 8);
    // ... the operand read of (wave row wm, quadrant row h, mt_l, ks) by lane l addresses row 128 wm + 64 h + 16 mt_l + (l & 15),
    //     channels 32 ks + 8 (l >> 4) of the K-tile, and every ds_read_b128 is bank-conflict free for the four lane groups
    for (int wm = 0; wm < 2; ++wm)
        for (int mt = 0; mt < 4; ++mt)
            for (int ks = 0; ks < 2; ++ks) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int a = a_frag_byte(wm, mt, ks, lane);
                    CHECK(a == a_frag_byte(wm, 0, ks, lane) + mt * 2048);                        // the kernel's immediate offsets
                    for (int h = 0; h < 2; ++h) {
                        const int rl = wm * 64 + mt * 16 + (lane & 15);
                        CHECK(a_tile_row(h, rl) == wm * 128 + h * 64 + mt * 16 + (lane & 15));
                        CHECK(a == a_half_byte(rl, ks * 4 + (lane >> 4)));
                    }
                }
                for (int g = 0; g < 4; ++g) {
                    std::set<int> banks;
                    for (int i = 0; i < 16; ++i) {
                        const int a = a_frag_byte(wm, mt, ks, groups[g][i]);
                        CHECK(a % 16 == 0 && a >= 0 && a + 16 <= HALF_BYTES);
                        for (int w = 0; w < 4; ++w) CHECK(banks.insert((a / 4 + w) % 64).second);
                    }
                }
            }
    // 3. weights: wave w requests tiles 2w, 2w + 1 of a half; the two halves hold every (16-column tile, k-tile) of the
    //    256-column block once; wave column wn reads tiles (wn * 2 + jq) * 2 + ks of half h = its columns 4 wn + 2 h + jq
    std::set<std::pair<int, int>> wseen;
    for (int h = 0; h < 2; ++h)
        for (int ti = 0; ti < 16; ++ti) {
            const int jt = b_tile_jt(h, ti), ks = b_tile_ks(ti);
            CHECK(jt >= 0 && jt < BJT && wseen.insert({jt, ks}).second);
            const int wn = ti >> 2, jq = (ti >> 1) & 1;
            CHECK(jt == wn * 4 + h * 2 + jq && b_half_byte(wn, jq, ks, 0) == ti * 1024);
        }
    CHECK((int)wseen.size() == BJT * 2);
    for (int wn = 0; wn < 4; ++wn)
        for (int jq = 0; jq < 2; ++jq)
            for (int ks = 0; ks < 2; ++ks) {
                CHECK(b_half_byte(wn, jq, ks, 63) + 16 <= HALF_BYTES);
                CHECK(b_half_byte(wn, jq, ks, 5) == b_half_byte(wn, 0, 0, 5) + (jq * 2 + ks) * 1024);   // the kernel's immediates
                for (int g = 0; g < 4; ++g) {
                    std::set<int> banks;
                    for (int i = 0; i < 16; ++i)
                        for (int w = 0; w < 4; ++w) CHECK(banks.insert((b_half_byte(wn, jq, ks, groups[g][i]) / 4 + w) % 64).second);
                }
            }
    // 4. the LDS-DMA schedule of the loop: replay the request / wait / read sequence of one wave for n K-tiles and check that
    //    (RAW) a half-tile is read only after a counted wait has retired its requests, one phase earlier at the least, and
    //    (WAR) a half-tile is re-requested only in a later phase than its last read (B0: one phase later, the others two)
    for (int n_ktiles = 2; n_ktiles <= 12; n_ktiles += 2) {
        struct Req { int slot, ktile, phase; };
        std::vector<Req> fifo;                                   // outstanding requests, oldest first (2 instructions each)
        std::map<int, int> landed_ktile, landed_phase, last_read_phase;   // per slot (stage * 4 + kind * 2 + h)
        int phase = 0;
        auto slot = [](int stage, int kind, int h) { return stage * 4 + kind * 2 + h; };
        auto request = [&](int kind, int h, int kt, int stage) {
            const int sl = slot(stage, kind, h);
            if (last_read_phase.count(sl)) { if (!(phase > last_read_phase[sl] + (kind == 1 && h == 0 ? 0 : 1))) return false; }
            fifo.push_back({sl, kt, phase});
            return true;
        };
        auto wait = [&](int keep_instr) {                        // s_waitcnt vmcnt(keep): all but the youngest keep / 2 half-tiles land
            while ((int)fifo.size() * 2 > keep_instr) { landed_ktile[fifo[0].slot] = fifo[0].ktile; landed_phase[fifo[0].slot] = phase; fifo.erase(fifo.begin()); }
        };
        auto read = [&](int kind, int h, int kt, int stage) {
            const int sl = slot(stage, kind, h);
            for (auto& r : fifo) if (r.slot == sl) return false;                       // a request into it still in flight
            if (!landed_ktile.count(sl) || landed_ktile[sl] != kt) return false;       // holds another K-tile
            if (!(landed_phase[sl] < phase)) return false;                             // the wait must be a phase earlier
            last_read_phase[sl] = phase;
            return true;
        };
        CHECK(request(1, 0, 0, 0) && request(0, 0, 0, 0) && request(1, 1, 0, 0) && request(0, 1, 0, 0));
        CHECK(request(1, 0, 1, 1) && request(0, 0, 1, 1) && request(1, 1, 1, 1));
        wait(6);
        for (int s = 0; s < n_ktiles; s += 2) {
            const bool last = s + 2 >= n_ktiles;
            ++phase; CHECK(read(1, 0, s, 0) && read(0, 0, s, 0)); CHECK(request(0, 1, s + 1, 1));
            ++phase; CHECK(read(1, 1, s, 0)); if (!last) CHECK(request(1, 0, s + 2, 0));
            ++phase; CHECK(read(0, 1, s, 0)); if (!last) CHECK(request(0, 0, s + 2, 0));
            ++phase; if (!last) { CHECK(request(1, 1, s + 2, 0)); wait(6); } else wait(0);
            ++phase; CHECK(read(1, 0, s + 1, 1) && read(0, 0, s + 1, 1)); if (!last) CHECK(request(0, 1, s + 2, 0));
            ++phase; CHECK(read(1, 1, s + 1, 1)); if (!last) CHECK(request(1, 0, s + 3, 1));
            ++phase; CHECK(read(0, 1, s + 1, 1)); if (!last) CHECK(request(0, 0, s + 3, 1));
            ++phase; if (!last) { CHECK(request(1, 1, s + 3, 1)); wait(6); }
        }
        CHECK(fifo.empty());
    }
    std::printf("ok\n");
    return 0;
}
"""


def test_gemm_8phase_index_arithmetic_and_schedule(tmp_path):
    """gemm_8phase.h (the K-loop of gemm_8phase_kernel): the half-tile images -- what the LDS-DMA requests deposit is what the
    operand reads expect, full coverage, bank-conflict-free ds_read_b128 -- and a replay of the loop's request / counted-wait /
    read schedule: no half-tile is read before a wait of an earlier phase has retired its requests, none is re-requested
    before the phase after its last read."""
    src = tmp_path / "harness8.cpp"
    src.write_text(HARNESS_8PHASE)
    exe = tmp_path / "harness8"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "jukebox_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
