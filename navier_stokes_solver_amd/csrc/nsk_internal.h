// nsk_internal.h — options of nsk_set_option that are NOT part of the public ABI (include/nsk.h):
// study switches kept for A/B measurements and the fault-injection hook of the tests.
#pragma once

enum {
  NSK_IOPT_TRI_X_LAYOUT = 6,    // blocked velocity factor: 2 (default) colour-ordered working vector when it runs
                                // single-launch, 0 the caller's order
  NSK_IOPT_FAULT_INJECT = 100,  // bit 0: the scalar (window) triangular solve walks its run list backwards,
                                // bit 1: the blocked velocity solve walks its upper half backwards — consumers before
                                // producers, so the bounded spins give up and the fallback has to take over
  NSK_IOPT_WINDOW_SPMV = 101,   // 1 (default): SpMV with S / Mp on the window format; 0: CSR-stream kernel
  NSK_IOPT_TINY_BYTES = 102     // triangular factors below this many bytes (default 4e6) are solved by ONE workgroup walking
                                // all levels; the tests set 0 to run the streamed kernels on small meshes
};
