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
