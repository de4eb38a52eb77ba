// Host (integer) half of prepare_inputs_labels_for_multimodal — llava/model/llava_arch.py:99-240.
//
// Bit-exact contract:
//   :144-145  padding stripped per row with attention_mask
//   :150-159  a row without IMAGE_TOKEN_INDEX still consumes one image-feature slot (zero-length slice)
//   :161-187  text pieces and image slots interleaved; image positions get label IGNORE_INDEX
//   :190-193  truncation to tokenizer_model_max_length happens AFTER image expansion
//   :196-225  pad to the batch max (left or right), attention mask, position_ids = arange on valid positions, 0 on pads
// Output `src` is the gather plan consumed by gather_embed_kernel: >= 0 token id, -1 zero row, -2-k image-feature row k.
#include <vector>

#include "engine.h"

namespace lmx {

static const int64_t kImageTokenIndex = -200;   // llava/constants.py:8
static const int64_t kIgnoreIndex = -100;       // llava/constants.py:7

int splice_plan(const int64_t* input_ids, const uint8_t* attention_mask, const int64_t* labels, int B, int L,
                int tokens_per_image, const int32_t* slot_rows, int n_image_slots, int max_len, int left_pad,
                int32_t* out_T, int32_t* src, uint8_t* out_mask, int64_t* out_pos, int64_t* out_labels) {
    LMX_REQUIRE(B > 0 && L > 0 && input_ids && out_T, "splice_plan: bad arguments");
    LMX_REQUIRE(tokens_per_image >= 0 && n_image_slots >= 0, "splice_plan: bad image geometry");
    // rows of the flattened feature matrix owned by each slot: uniform (4-D `images`) or per-slot (list / 5-D `images`,
    // llava_arch.py:114-119 flattens each entry to [n_i * P, H])
    std::vector<int> slot_base(n_image_slots + 1, 0);
    for (int k = 0; k < n_image_slots; ++k) {
        const int rows = slot_rows ? slot_rows[k] : tokens_per_image;
        LMX_REQUIRE(rows >= 0, "splice_plan: negative slot size");
        slot_base[k + 1] = slot_base[k] + rows;
    }
    std::vector<std::vector<int32_t>> rows_src(B);
    std::vector<std::vector<int64_t>> rows_lab(B);
    int cur_image_idx = 0;
    for (int b = 0; b < B; ++b) {
        std::vector<int32_t>& rs = rows_src[b];
        std::vector<int64_t>& rl = rows_lab[b];
        int n_img = 0;
        for (int i = 0; i < L; ++i) {
            if (attention_mask && !attention_mask[(size_t)b * L + i]) continue;
            if (input_ids[(size_t)b * L + i] == kImageTokenIndex) ++n_img;
        }
        if (n_img == 0) {
            // reference indexes image_features[cur_image_idx] even for a text-only row (IndexError if exhausted)
            LMX_REQUIRE(cur_image_idx < n_image_slots, "splice_plan: image_features index out of range (text-only row consumes a slot)");
            ++cur_image_idx;
        }
        for (int i = 0; i < L; ++i) {
            if (attention_mask && !attention_mask[(size_t)b * L + i]) continue;
            const int64_t id = input_ids[(size_t)b * L + i];
            if (id == kImageTokenIndex) {
                LMX_REQUIRE(cur_image_idx < n_image_slots, "splice_plan: more <image> markers than image features");
                const int base = slot_base[cur_image_idx], cnt = slot_base[cur_image_idx + 1] - base;
                for (int p = 0; p < cnt; ++p) { rs.push_back(-2 - (base + p)); rl.push_back(kIgnoreIndex); }
                ++cur_image_idx;
            } else {
                LMX_REQUIRE(id >= 0 && id <= 0x7fffffff, "splice_plan: token id out of range");
                rs.push_back((int32_t)id);
                rl.push_back(labels ? labels[(size_t)b * L + i] : kIgnoreIndex);
            }
        }
        if (max_len > 0 && (int)rs.size() > max_len) { rs.resize(max_len); rl.resize(max_len); }
    }
    int T = 0;
    for (int b = 0; b < B; ++b) T = (int)rows_src[b].size() > T ? (int)rows_src[b].size() : T;
    *out_T = T;
    if (!src) return 0;       // sizing call
    LMX_REQUIRE(out_mask && out_pos && out_labels, "splice_plan: output buffers missing");
    for (int b = 0; b < B; ++b) {
        const int n = (int)rows_src[b].size();
        const int off = left_pad ? T - n : 0;
        for (int t = 0; t < T; ++t) {
            const size_t o = (size_t)b * T + t;
            src[o] = -1; out_mask[o] = 0; out_pos[o] = 0; out_labels[o] = kIgnoreIndex;
        }
        for (int i = 0; i < n; ++i) {
            const size_t o = (size_t)b * T + off + i;
            src[o] = rows_src[b][i]; out_mask[o] = 1; out_pos[o] = i; out_labels[o] = rows_lab[b][i];
        }
    }
    return 0;
}

}  // namespace lmx
