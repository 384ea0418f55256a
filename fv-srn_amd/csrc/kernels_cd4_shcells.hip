#define FVSRN_CD 4
#define FVSRN_PART 6
#include "kernels_inst.inc"
