#define FVSRN_CD 8
#define FVSRN_PART 5
#include "kernels_inst.inc"
