// The library's A/B switches, read from the environment ONCE per process (crk_sw(), conv_kernels.hip).  None is needed in
// normal use: each selects an older generation of a kernel that stays in the library as the reference of a bitwise test
// (tests/test_gpu_properties.py), or a shape parameter of a measurement (tools/).  Defaults in brackets.
//   CRK_SK_V [2]        1: frame-split forward kernels without the folds (stack_fwd_kernel)
//   CRK_SKB_V [2]       1: frame-split data-gradient chain (stack_bwd_kernel)
//   CRK_PS_V [2]        1: frame-split plain chains (pstack_kernel)
//   CRK_NO_FUSE [0]     1: one kernel per layer (conv_tile_kernel / wgrad_kernel)
//   CRK_S2X [1]         0: bf16x3f forward on the frame-split split-operand kernels
//   CRK_DISC_SPLIT [1]  0: the discriminator's data-gradient chain frame-split
//   CRK_S2_CFG [0]      <ft><fh>[<ft1>]: pins stack2_fwd_kernel's window shape
//   CRK_SK_NW [0]       4|6|8 (two digits: forward, backward): window shape of the frame-split kernels
//   CRK_PS_NW [0]       4|8: window shape of pstack_kernel
//   CRK_WG_GROUPS [32]  partial-sum groups of the gated convs' weight gradients
//   CRK_WG_CPG [0]      64-frame chunks per weight-gradient group of the plain convs (0: derived)
//   CRK_WG_FILL [1]     0: gated stacks keep CRK_WG_GROUPS utterance groups however few blocks they have (1: as many
//                       64-frame-chunk groups as put one workgroup per (group, block) on every compute unit)
//   CRK_VQ_F16 [1]      0: the exact fp32-MFMA codebook search
//   CRK_VQ_LC [2]       0 frame-per-lane, 1 code-per-lane, 2 MFMA search
//   CRK_LOGMEL_WAVE [1] 0: radix-2 log-mel kernel (one workgroup per frame)
#ifndef CRK_SWITCHES_H
#define CRK_SWITCHES_H
struct CrkSwitches {
  int sk_v, skb_v, ps_v, no_fuse, s2x, disc_split, s2_cfg, sk_nw_fwd, sk_nw_bwd, ps_nw, wg_groups, wg_cpg, wg_fill, vq_f16, vq_lc, logmel_wave;
};
const CrkSwitches& crk_sw();
#endif
