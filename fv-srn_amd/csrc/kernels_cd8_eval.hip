#define FVSRN_CD 8
#define FVSRN_PART 0
#include "kernels_inst.inc"
