/* oracle/hip_host/mfma_stubs.c -- TEST INFRASTRUCTURE ONLY.  ext_comm.hip (RCCL) is not part of the host emulation: the one entry point the
 * C++ runner references directly refuses, so a tensor-parallel runner reports the failure instead of crashing. */
int mrs_comm_all_reduce_sum_f32() { return -1; }
