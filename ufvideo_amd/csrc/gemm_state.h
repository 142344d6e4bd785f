// Per-device state of the split-K / stream-K GEMM forms (gemm_state.hip): turn-flag ring, ticket bases, the timeout word.
#pragma once
int ufv_dev_n_cu();                                                                     // CUs of the current device (cached per device)
int ufv_splitk_acquire(int tiles, int parts, int** flags, int* base, int** err);        // a private flag slice + ticket base for ONE launch
int ufv_streamk_acquire(int grid, float** ws, int** flags, int* epoch, int** err);      // opt-in stream-K: one launch at a time per device
