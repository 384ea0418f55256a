#define FVSRN_CD 2
#define FVSRN_PART 0
#include "kernels_inst.inc"
