/* stands in for OpenBLAS's openblas_config.h (src/gemma.cpp:79 prints OPENBLAS_VERSION): the library used is the
   OpenBLAS inside scipy, reached through blas_bridge.c */
#define OPENBLAS_VERSION " OpenBLAS (scipy build) "
