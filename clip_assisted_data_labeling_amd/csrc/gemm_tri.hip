// Near-duplicate search GEMM (K11, /root/reference/_2_remove_duplicates.py:69-80) on the persistent pipeline of
// gemm_persist.hip: f16 operands, upper-triangular tile list, threshold + atomic append epilogue.  A translation unit of its
// own (see the note at ce_gemm_tri_persist in gemm_persist.hip).
#define GEMM_PERSIST_TRI_TU 1
#include "gemm_persist.hip"
