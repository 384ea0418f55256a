#define FVSRN_CD 4
#define FVSRN_PART 3
#include "kernels_inst.inc"
