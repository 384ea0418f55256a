// activation template indices shared by host and device code
#pragma once
namespace fvsrn {
// ACT_RELU01: ReLU on the [0,1]-scaled weight image (pack.cpp): convert+ReLU is one clamped v_cvt_pk_f16_f32
enum { ACT_RELU = 0, ACT_SINE = 1, ACT_SNAKE = 2, ACT_SNAKEALT = 3, ACT_RELU01 = 4, ACT_SIGMOID = 5 };
}
