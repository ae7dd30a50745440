// examples/dump_vectors.rs -- NOT part of the MI355X build.  Copy into a checkout of EricLBuehler/candle-vllm and run
//   cargo run --release --example dump_vectors -- ref_inputs.json ref_outputs.json
// It pushes the explicit inputs of tests/golden/ref_inputs.json (written by oracle/ref_vectors/make_inputs.py) through the reference's
// own CPU arithmetic (guoqingbao/candle rev cafd231 as pinned by Cargo.toml:24-25) at the call sites of the decode hot path and writes
// the results as JSON, so that tests/test_cpu_ref_vectors.py can pin the MI355X repository's oracle to the reference itself.
// Written against candle 0.8's public API without a compiler (no Rust toolchain where it was written): expect to fix a signature or two.
use candle_core::quantized::{GgmlDType, QMatMul, QStorage, QTensor};
use candle_core::{DType, Device, Module, Result, Tensor, D};
use serde_json::{json, Map, Value};
use std::borrow::Cow;

fn f32s(v: &Value) -> Vec<f32> {
    v.as_array().unwrap().iter().map(|x| x.as_f64().unwrap() as f32).collect()
}
fn dims(v: &Value) -> Vec<usize> {
    v.as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect()
}
fn hex(v: &Value) -> Vec<u8> {
    let s = v.as_str().unwrap().as_bytes();
    (0..s.len() / 2)
        .map(|i| u8::from_str_radix(std::str::from_utf8(&s[2 * i..2 * i + 2]).unwrap(), 16).unwrap())
        .collect()
}
fn tensor(v: &Value, dev: &Device) -> Result<Tensor> {
    Tensor::from_vec(f32s(&v["data"]), dims(&v["shape"]), dev)
}
fn out(t: &Tensor) -> Result<Value> {
    let t = t.to_dtype(DType::F32)?.flatten_all()?;
    Ok(json!(t.to_vec1::<f32>()?))
}

fn main() -> Result<()> {
    let args: Vec<String> = std::env::args().collect();
    let inp: Value = serde_json::from_str(&std::fs::read_to_string(&args[1]).unwrap()).unwrap();
    let dev = Device::Cpu;
    let mut res = Map::new();

    // ---- QMatMul::forward on CPU (linear.rs:765-806: x is F32; candle quantises it to Q8_K per 256 and takes integer dots)
    for case in inp["qmatmul"].as_array().unwrap() {
        let dtype = match case["ggml_type"].as_str().unwrap() {
            "q4_k" => GgmlDType::Q4K,
            "q6_k" => GgmlDType::Q6K,
            "q8_0" => GgmlDType::Q8_0,
            other => panic!("unknown type {other}"),
        };
        let (n, k) = (case["n"].as_u64().unwrap() as usize, case["k"].as_u64().unwrap() as usize);
        let storage = QStorage::from_data(Cow::Owned(hex(&case["blocks_hex"])), &dev, dtype)?;
        let w = QTensor::new(storage, (n, k))?;
        let deq = w.dequantize(&dev)?; // what a dequantise-then-matmul path sees (the oracle's O1 weights)
        let mm = QMatMul::from_qtensor(w)?;
        let x = tensor(&case["x"], &dev)?;
        let y = mm.forward(&x)?;
        res.insert(
            case["name"].as_str().unwrap().to_string(),
            json!({"y": out(&y)?, "dequantized": out(&deq)?, "y_dequant_matmul": out(&x.matmul(&deq.t()?)?)?}),
        );
    }
    // ---- rms_norm (layers/qrmsnorm.rs:28-31), silu * up (quantized_llama.rs:33-37)
    for case in inp["rms_norm"].as_array().unwrap() {
        let x = tensor(&case["x"], &dev)?;
        let w = tensor(&case["w"], &dev)?;
        let y = candle_nn::ops::rms_norm(&x, &w, case["eps"].as_f64().unwrap() as f32)?;
        res.insert(case["name"].as_str().unwrap().to_string(), json!({"y": out(&y)?}));
    }
    for case in inp["silu_mul"].as_array().unwrap() {
        let g = tensor(&case["gate"], &dev)?;
        let u = tensor(&case["up"], &dev)?;
        let y = (candle_nn::ops::silu(&g)? * u)?;
        res.insert(case["name"].as_str().unwrap().to_string(), json!({"y": out(&y)?}));
    }
    // ---- RoPE (layers/rotary_emb.rs:72-100): xs [b, h, t, d], cos / sin [t, d/2]
    for case in inp["rope"].as_array().unwrap() {
        let x = tensor(&case["x"], &dev)?;
        let cos = tensor(&case["cos"], &dev)?;
        let sin = tensor(&case["sin"], &dev)?;
        let y = if case["interleaved"].as_bool().unwrap() {
            candle_nn::rotary_emb::rope_i(&x, &cos, &sin)?
        } else {
            candle_nn::rotary_emb::rope(&x, &cos, &sin)?
        };
        res.insert(case["name"].as_str().unwrap().to_string(), json!({"y": out(&y)?}));
    }
    // ---- NaiveAttention::forward (models/mod.rs:1288-1306) on BF16 tensors: q [1, H, 1, D], k / v [1, Hkv, T, D] (repeat_kv :1240-1247)
    for case in inp["attention_bf16"].as_array().unwrap() {
        let q = tensor(&case["q"], &dev)?.to_dtype(DType::BF16)?;
        let k = tensor(&case["k"], &dev)?.to_dtype(DType::BF16)?;
        let v = tensor(&case["v"], &dev)?.to_dtype(DType::BF16)?;
        let n_rep = case["n_rep"].as_u64().unwrap() as usize;
        let rep = |x: &Tensor| -> Result<Tensor> {
            if n_rep == 1 {
                return Ok(x.clone());
            }
            let (b, hkv, t, d) = x.dims4()?;
            Tensor::cat(&vec![x; n_rep], 2)?.reshape((b, hkv * n_rep, t, d))
        };
        let (k, v) = (rep(&k)?.contiguous()?, rep(&v)?.contiguous()?);
        let scale = case["scale"].as_f64().unwrap();
        let w = (q.matmul(&k.transpose(2, 3)?)? * scale)?;
        let p = candle_nn::ops::softmax_last_dim(&w)?;
        let o = p.matmul(&v)?;
        res.insert(
            case["name"].as_str().unwrap().to_string(),
            json!({"scores": out(&w)?, "probabilities": out(&p)?, "y": out(&o)?}),
        );
    }
    // ---- greedy sampling (logits_processor.rs:92-95): which index wins a tie?
    for case in inp["argmax"].as_array().unwrap() {
        let x = tensor(&case["x"], &dev)?;
        let idx = x.argmax(D::Minus1)?.to_dtype(DType::U32)?;
        res.insert(case["name"].as_str().unwrap().to_string(), json!({"index": idx.flatten_all()?.to_vec1::<u32>()?}));
    }
    std::fs::write(&args[2], serde_json::to_string(&Value::Object(res)).unwrap()).unwrap();
    Ok(())
}
