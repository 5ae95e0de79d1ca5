// Index arithmetic of gemm_glds_kernel (gemm.hip; stand-alone form: tools/gemm_glds_probe.hip): 128 x 128 output tile, 64
// channels per K-step, both operands brought into LDS by LDS-DMA (`global_load_lds_dwordx4`).  Kept apart from the kernel so that the CPU
// suite can check it exhaustively (tests/test_gemm_glds_index.py compiles this header with g++).
//
// An LDS-DMA instruction writes lane i's 16 bytes to  base + 16 * i  -- the LDS image is lane-linear per instruction, only
// the SOURCE address is per lane (cdna_hip_programming.md section 5, "glds K-tile trap").  So:
//   activations: one instruction moves 8 rows x 128 bytes (full lines).  Lane i fetches row (i >> 3), segment
//     (i & 7) ^ (i >> 3) of the line: in LDS the segments of row r sit XOR-ed with r, and the MFMA operand read
//     (16 rows x one segment per 16-lane group) touches every bank once.
//   weights: the packed image (gemm.hip: P[jt][kt][lane][8]) already is the operand order; one instruction moves one
//     16-column x 32-channel tile, lane i fetches its own fragment.
#pragma once

#if defined(__HIPCC__)
#define GI_FN __host__ __device__ inline
#else
#define GI_FN inline
#endif

namespace gi {
constexpr int BM = 128;                 // rows of the output tile
constexpr int BJT = 8;                  // 16-column tiles of the output tile (128 columns)
constexpr int KSTEP = 64;               // channels per K-step = two 32-channel k-tiles of the packed weights
constexpr int A_PIECES = BM / 8;        // LDS-DMA instructions per K-step for the activation panel
constexpr int W_TILES = BJT * 2;        // ... for the weight panel
constexpr int A_BYTES = A_PIECES * 1024;
constexpr int STAGE_BYTES = A_BYTES + W_TILES * 1024;      // 32 KiB

// LDS-DMA instruction `piece` of the activation panel: which row of the piece / which 16-byte segment of its 128-byte
// line lane `lane` fetches (it lands at byte  piece * 1024 + lane * 16  of the stage).
GI_FN int a_src_row(int lane) { return lane >> 3; }
GI_FN int a_src_seg(int lane) { return (lane & 7) ^ (lane >> 3); }
// Byte offset in the stage of segment `seg` (0..7: channels seg*8 .. seg*8+7 of the K-step) of tile row `row` (0..127).
GI_FN int a_byte(int row, int seg) { return (row >> 3) * 1024 + ((((row & 7) << 3) | (seg ^ (row & 7))) << 4); }
// Weight tile (jt, ks) of the stage (jt: 16-column tile 0..7, ks: k-tile 0..1 of the K-step), lane's fragment.
GI_FN int w_byte(int jt, int ks, int lane) { return A_BYTES + ((jt * 2 + ks) << 10) + (lane << 4); }
// MFMA 16x16x32 operand of the activations: lane l holds row (l & 15) of the 16-row tile, channels (l >> 4)*8 .. +7 of the
// 32-channel k-tile  ->  row of the 128-row tile, segment of the K-step.
GI_FN int frag_row(int wave_m, int mt, int lane) { return wave_m * 64 + mt * 16 + (lane & 15); }
GI_FN int frag_seg(int ks, int lane) { return ks * 4 + (lane >> 4); }

// XCD-aware order of the output tiles: block b runs on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch); an XCD keeps whole
// row panels -- all column tiles of a 128-row panel are consecutive in ITS sequence, so the panel's activations are fetched
// from HBM once, by that XCD's L2.  Returns false for the padding blocks of a grid of  ceil(MB / 8) * 8 * NB  blocks.
GI_FN bool tile_of_block(int b, int MB, int NB, int* mp, int* nt) {
    const int xcd = b & 7, q = b >> 3;
    *mp = (q / NB) * 8 + xcd;
    *nt = q % NB;
    return *mp < MB;
}
}  // namespace gi
