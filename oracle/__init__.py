"""CPU oracle for the SpeechCLIP+ contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``speechclip_plus_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and there only as the checker / the timed CPU baseline.

It is a plain torch-CPU fp32 restatement of the reference's arithmetic:

* ``hubert_ref``   - fairseq HuBERT as patched by avssl/module/speech_encoder_plus.py:29-107
                     (third-party arithmetic: fairseq @ b5a039c2, requirements.txt:6)
* ``head_ref``     - WeightedSumLayer, get_keypadding_mask, KW_ParallelBranch,
                     TransformerEncoder / MultiheadAttentionAndNorm
* ``loss_ref``     - MaskedContrastiveLoss (avssl/module/losses.py:129-245)
* ``retrieval_ref``- mutualRetrieval (avssl/module/retrieval.py:6-121)

Pinning status (see DESIGN.md "Oracle"):
* head / loss / weighted-sum / mask / retrieval: pinned against the reference's own leaf
  files, imported by path in the build container (tests/golden/make_golden.py), outputs
  committed as fixtures in tests/golden/*.npz.
* HuBERT (fairseq is absent from /root/reference and from the image): cross-checked
  against the independent ``transformers.HubertModel`` built from a local config;
  no reference test pins numbers at the fairseq boundary => "parity unpinned" for the
  encoder arithmetic itself, pinned only to that independent implementation.
"""
from .lengths import (conv_out_lengths, fairseq_valid_frames, feat_len_rule,
                      get_keypadding_mask)
from .hubert_ref import (HubertArch, hubert_forward, init_hubert_weights, speech_encoder_forward, bf16_store,
                         bf16_weights, SitedStore)
from .head_ref import (weighted_sum, parallel_branch_forward, init_parallel_branch_weights,
                       transformer_encoder_forward, mha_and_norm_forward)
from .loss_ref import masked_contrastive_loss
from .retrieval_ref import mutual_retrieval
from .cascaded_ref import (cascaded_plus_forward, hybrid_plus_forward, cif_forward, vq_forward, clip_encode_keywords,
                           clip_text_transformer)
