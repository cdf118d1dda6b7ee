//! Host `Vec<f32>` weights / state -> HBM, and the views over them, for the `hip` feature.
use crate::device::hip::{Hip, HipSlice};
use super::state::{RunState, TransformerWeights, TransformerWeightsView};
use super::View;

macro_rules! upload {
    ($dev:expr, $src:expr, { $($f:ident),* $(,)? } $(, $extra:ident : $val:expr)*) => {
        Self { $($f: $dev.allocate(&$src.$f),)* $($extra: $val,)* }
    };
}

impl RunState<HipSlice> {
    /// zero-filled host state (ram.rs:7-23) -> device, one allocation per buffer
    pub fn from_state(state: &mut RunState<Vec<f32>>, device: &Hip) -> Self {
        upload!(device, state, { x, xb, xb2, hb, hb2, q, k, v, att, logits, key_cache, value_cache })
    }
}

impl TransformerWeights<HipSlice> {
    pub fn from_weight(tw: &mut TransformerWeights<Vec<f32>>, device: &Hip) -> Self {
        upload!(device, tw, { token_embedding_table, rms_att_weight, rms_ffn_weight, wq, wk, wv, wo, w1, w2, w3,
                              rms_final_weight, freq_cis_real, freq_cis_imag, wcls },
                wcls_exists: tw.wcls_exists)
    }
}

impl<'a> TransformerWeightsView<'a, HipSlice> {
    pub fn from_hip_ws(ws: &'a TransformerWeights<HipSlice>) -> Self {
        macro_rules! view { ($($f:ident),*) => { TransformerWeightsView {
            $($f: View::new(&ws.$f),)*
            // a tied classifier reads the embedding table (state.rs:111-117); `wcls` is then a 1-element placeholder
            wcls: if ws.wcls_exists { View::new(&ws.wcls) } else { View::new(&ws.token_embedding_table) },
            wcls_exists: ws.wcls_exists,
        } } }
        view!(token_embedding_table, rms_att_weight, rms_ffn_weight, wq, wk, wv, wo, w1, w2, w3,
              rms_final_weight, freq_cis_real, freq_cis_imag)
    }
}

// RunStateView::from_rs (state.rs:35-50) is generic over the storage and needs no HIP twin.
