#define FVSRN_CD 6
#include "kernels_inst.inc"
