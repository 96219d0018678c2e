// knobs.h — the library's launch-plan switches, resolved ONCE.
//
// Every AMX_* environment variable the library understands is a row of the table in knobs.hip.  The table is read on the
// first call that needs a plan (under a mutex) into an immutable AmxKnobs that every later launch reads through one
// atomic pointer load: no getenv on the launch path, and two host threads launching on two streams see the same plan
// (include/atomai_amd.h promises re-entrancy across streams).  amx_knobs_reload() re-reads the environment — the hook
// of in-process A/B scripts and of the tests that compare two plans; product code never calls it.
#pragma once

struct AmxKnobs {
    int conv_lattice;     // AMX_CONV_LATTICE    1: dilations 2/4/6 as d*d plain 3x3 convolutions; 0: halo-class kernels
    int conv_nt;          // AMX_CONV_NT         0: plan_conv's choice; 1|2|4: cout tiles of 16 per workgroup
    int conv_th;          // AMX_CONV_TH         0: plan_conv's choice; 8|16: tile rows
    int conv_rem;         // AMX_CONV_REM        1: 28 / 52 stored channels as 16+3x4 / 3x16+4 columns; 0: padded 32 / 2x32
    int conv_xcd;         // AMX_CONV_XCD        0 off, 1 all (default), 2 launches with > 1 cout block, 3 dilated launches only
    int conv_xpack;       // AMX_CONV_XPACK      1: lattice sub-images of one residue row share a tile axis when that saves tiles
    int bwd_fuse;         // AMX_BWD_FUSE        1: BatchNorm / LeakyReLU backward inside the consumers' loaders
    int bwd_sums;         // AMX_BWD_SUMS        1: BatchNorm-backward sums of the source layer in the data-gradient epilogue
    int conv_ws;          // AMX_CONV_WS         0 off, 1 default classes, 2 forward only, 3 data gradients only
    int conv_ws_dgrad;    // AMX_CONV_WS_DGRAD   bit mask of data-gradient classes on the wave-specialised kernel
    int wgrad_th;         // AMX_WGRAD_TH        0: plan_wgrad's choice; 4|8: tile rows
    int wgrad_wgs;        // AMX_WGRAD_WGS       0: one split-K workgroup per CU; n: target workgroup count (tests)
    int wgrad_ws;         // AMX_WGRAD_WS        0: weight gradients on wgrad_kernel.h only
    int wgrad_ws_mask;    // AMX_WGRAD_WS_MASK   wave layouts on wgrad_ws: 1 = 16, 2 = 32, 4 = >= 64 input channels
    int wgrad_ws_wm4;     // AMX_WGRAD_WS_WM4    64-channel class: h > 0 at least h rows, h < 0 at most -h rows, 0 never
    int gemm_tile;        // AMX_GEMM_TILE       0: by workgroup count; 32|64
    int rdec_fwd_mt;      // AMX_RDEC_FWD_MT     pixels per tile of the 128-unit rDecoder forward kernel (64|128)
    int rdec_bwd_mt;      // AMX_RDEC_BWD_MT     pixels per tile of the 128-unit rDecoder backward kernel (64|32)
};

// The frozen plan switches (first call resolves them; lock-free afterwards).
const AmxKnobs& amx_knobs();
