#define FVSRN_CD 8
#define FVSRN_PART 6
#include "kernels_inst.inc"
