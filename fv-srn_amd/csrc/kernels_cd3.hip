#define FVSRN_CD 3
#include "kernels_inst.inc"
