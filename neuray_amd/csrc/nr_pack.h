// Host-side packing of the reference state_dict tensors into MFMA A-fragments (see nr_layout.h).
#pragma once
#include <cstddef>
#include <vector>
#include "nr_layout.h"

namespace nr {

struct LayerMaps {
    std::vector<int> out_map;   // [mt_out*16]  MFMA row m of tile mo -> weight row (or -1)
    std::vector<int> in_map;    // [kq*4][4]    (k-step, lane group) -> weight column (or -1)
    std::vector<int> in1_map;   // [k1][4]
};

// Writes layer `layer` (quads, singles, bias) at its offset inside `dst` (kPackedPassFloats floats).
void pack_layer(float* dst, int layer, const float* W, int ldw, const float* bias, const LayerMaps& maps);

// Writes the vector rows of `layer` (nr_layout.h kVec): row j = weight row rows[j], features = columns col0.. of W.
void pack_vec(float* dst, int layer, const float* W, int ldw, const float* bias, const int* rows, int col0, int nfeat);

// tensors: array of T_COUNT host pointers in PassTensor order (vis-head entries may be null).
// Returns 0 on success, non-zero if a required tensor is missing.
// fold: prob_embed.2 (a Linear with no activation behind it) is multiplied into its consumers neuray_fc.0 and base_fc.0's last 32
// columns (aggregate_net.py:27-31, ibrnet.py:337,342-343); the L_PE2 slot stays zero and the kernel skips the layer.
int pack_pass_weights(const float* const* tensors, float* dst, bool fold = false);

// AR_X3 (nr_layout.h): the folded pack with every quad weight split into three bf16 parts, in the pair layout of the K = 32 bf16
// MFMA (kPackedPointFloatsX3 floats: the point kernel's layers only; the ray kernel keeps reading the fp32 pack).  Inference only.
int pack_pass_weights_x3(const float* const* tensors, float* dst);
// w = part[0] + part[1] + part[2] exactly (bf16 bit patterns, each the round-to-nearest of what the parts before it left)
void split3_bf16(float w, unsigned short (&part)[3]);

// packed[i] = flat[index[i]] * scale[i] (flat natural layout, nr_layout.h); index -1 = padding.  kPackedPassFloats entries each.
int pack_pass_index_map(bool has_vis, int* index, float* scale);

// The transposed layers of the backward pass (nr_layout.h LT_*) -> dst (kPackedTFloats floats); true weights.
int pack_pass_t_weights(const float* const* tensors, float* dst);
// packed_t[i] = flat[index[i]]; index -1 = padding.  kPackedTFloats entries.
int pack_pass_t_index_map(bool has_vis, int* index);

// np.random.shuffle (legacy MT19937 RandomState) of a 1-D array of 4- or 8-byte items, in place; key[624] / *pos are the generator state
// and are advanced exactly as numpy advances them.  0 on success.
int mt19937_shuffle(unsigned int* key, int* pos, void* data, long long n, int itemsize);

}  // namespace nr
