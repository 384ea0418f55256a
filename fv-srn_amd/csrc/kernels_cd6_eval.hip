#define FVSRN_CD 6
#define FVSRN_PART 0
#include "kernels_inst.inc"
