#define FVSRN_CD 3
#define FVSRN_PART 6
#include "kernels_inst.inc"
