#define FVSRN_CD 4
#define FVSRN_PART 2
#include "kernels_inst.inc"
