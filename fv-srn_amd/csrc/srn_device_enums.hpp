// activation template indices shared by host and device code
#pragma once
namespace fvsrn {
// Two of them exist only on a second, re-scaled weight image that the renderer uses where it is exact (pack.cpp):
// ACT_RELU01:    ReLU with all activations scaled into [0,1] by powers of two: convert + ReLU is one clamped v_cvt_pk_f16_f32
// ACT_SNAKEALT0: SnakeAlt with parameter p = 2^k: (x + 1 - cos(2 p x)) / (2p) = b (x - c) + b with b = 2^-(k+1); the factor b goes
//                into the next layer's weights (exact) and b * sum(W) into its fp32 bias, leaving x - cos(2 p x) to the VALU
enum { ACT_RELU = 0, ACT_SINE = 1, ACT_SNAKE = 2, ACT_SNAKEALT = 3, ACT_RELU01 = 4, ACT_SIGMOID = 5, ACT_SNAKEALT0 = 6 };
}
