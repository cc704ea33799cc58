/* TEST / BASELINE INFRASTRUCTURE -- not part of the product path (only tests/, bench.py's cpu_baseline leg and __graft_entry__ may use oracle/).
 *
 * Greedy hard NMS per class segment, float32 arithmetic in the operation order of ssd_oracle.nms_single_class (the restatement of
 * torchvision.ops.nms, called by the reference at demonet/models/generalized_ssd.py:389 through batched_nms):
 *   area = (x2 - x1) * (y2 - y1);  inter = max(0, min(x2) - max(x1)) * max(0, min(y2) - max(y1));
 *   IoU = inter / (area_i + area_j - inter);  j is suppressed when IoU > thr (strict).
 * The candidates of a segment must already be sorted by score, descending (that is how select_candidates / torch.topk emit them).
 * This is what a compiled CPU NMS (torchvision's C++ kernel) costs, for the cpu_baseline leg of bench.py: the checker itself
 * (ssd_oracle.postprocess_detections) stays the numpy / Python loop.
 *
 * build: gcc -O2 -fPIC -shared -ffp-contract=off -o _build/libnms_c.so nms_c.c   (done by __graft_entry__.build() / fast_post.py)
 */
#include <stdint.h>

int nms_segments(const float* boxes, const int32_t* seg_start, int nseg, float thr, uint8_t* keep) {
    int kept = 0;
    for (int s = 0; s < nseg; ++s) {
        const int b = seg_start[s], e = seg_start[s + 1];
        for (int i = b; i < e; ++i) keep[i] = 1;
        for (int i = b; i < e; ++i) {
            if (!keep[i]) continue;
            ++kept;
            const float x1 = boxes[4 * i], y1 = boxes[4 * i + 1], x2 = boxes[4 * i + 2], y2 = boxes[4 * i + 3];
            const float ai = (x2 - x1) * (y2 - y1);
            for (int j = i + 1; j < e; ++j) {
                if (!keep[j]) continue;
                const float u1 = boxes[4 * j], v1 = boxes[4 * j + 1], u2 = boxes[4 * j + 2], v2 = boxes[4 * j + 3];
                const float aj = (u2 - u1) * (v2 - v1);
                const float xx1 = x1 > u1 ? x1 : u1, yy1 = y1 > v1 ? y1 : v1;
                const float xx2 = x2 < u2 ? x2 : u2, yy2 = y2 < v2 ? y2 : v2;
                float w = xx2 - xx1, h = yy2 - yy1;
                if (w < 0.0f) w = 0.0f;
                if (h < 0.0f) h = 0.0f;
                const float inter = w * h;
                const float ovr = inter / (ai + aj - inter);
                if (ovr > thr) keep[j] = 0;
            }
        }
    }
    return kept;
}
