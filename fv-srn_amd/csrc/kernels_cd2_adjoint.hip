#define FVSRN_CD 2
#define FVSRN_PART 4
#include "kernels_inst.inc"
