#define FVSRN_CD 3
#define FVSRN_PART 4
#include "kernels_inst.inc"
