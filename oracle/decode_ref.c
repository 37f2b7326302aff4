/*
 * oracle/decode_ref.c -- CPU ORACLE (test infrastructure, not the product).
 *
 * One level of the head's decode in plain C, op for op (fp32, no contraction):
 *   mmdet/models/dense_heads/yolocsp_head.py:263-285   sigmoid, xy = 2s-1, wh = (2s)^2
 *   mmdet/core/anchor/anchor_generator.py:255-270       anchor = base + shift
 *   mmdet/core/bbox/coder/yolov4_bbox_coder.py:51-66    centre/size -> corners
 * Used to cross-check the torch statement in yolov4_oracle.decode_maps; sigmoid uses
 * 1/(1+expf(-x)).
 *
 * pred: one image, NHWC (H, W, A*(5+C)); base: A x 4; outputs for H*W*A boxes.
 */
#include <math.h>

void oracle_decode_level(const float* pred, int H, int W, int A, int C, int stride, int rescale_unused,
                         const float* base, float* boxes, float* conf, float* cls, float* unused) {
  (void)rescale_unused; (void)unused;
  const int attr = 5 + C;
  const float fs = (float)stride;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int a = 0; a < A; ++a) {
        const int j = (y * W + x) * A + a;
        const float* p = pred + (long)j * attr;
        float s[5];
        for (int k = 0; k < 5; ++k) s[k] = 1.f / (1.f + expf(-p[k]));
        const float sx = (float)(x * stride), sy = (float)(y * stride);
        const float ax1 = base[4 * a + 0] + sx, ay1 = base[4 * a + 1] + sy;
        const float ax2 = base[4 * a + 2] + sx, ay2 = base[4 * a + 3] + sy;
        const float px = s[0] * 2.f - 1.f, py = s[1] * 2.f - 1.f;
        const float tw = s[2] * 2.f, th = s[3] * 2.f;
        const float pw = tw * tw, ph = th * th;
        const float xc = (ax1 + ax2) * 0.5f, yc = (ay1 + ay2) * 0.5f;
        const float aw = ax2 - ax1, ah = ay2 - ay1;
        const float xcp = px * fs + xc, ycp = py * fs + yc;
        const float wp = pw * aw, hp = ph * ah;
        boxes[4 * j + 0] = xcp - wp / 2.f;
        boxes[4 * j + 1] = ycp - hp / 2.f;
        boxes[4 * j + 2] = xcp + wp / 2.f;
        boxes[4 * j + 3] = ycp + hp / 2.f;
        conf[j] = s[4];
        for (int c = 0; c < C; ++c) cls[(long)j * C + c] = 1.f / (1.f + expf(-p[5 + c]));
      }
}
