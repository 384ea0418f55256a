#define FVSRN_CD 8
#include "kernels_inst.inc"
