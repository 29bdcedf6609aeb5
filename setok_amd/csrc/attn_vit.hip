// attn_vit.hip — bf16 MFMA attention for the ViT tower shape (uniform segments, head dim 64).
#include "common.h"

int setok_attention_vit_bf16(hipStream_t s, const bf16* qkv, bf16* out, int n_imgs, int T, int H, int Dh, float scale) {
    (void)s; (void)qkv; (void)out; (void)n_imgs; (void)T; (void)H; (void)Dh; (void)scale;
    return SETOK_EUNSUPPORTED;   // falls back to the generic kernel (norm_attn.hip) until the MFMA kernel lands
}
