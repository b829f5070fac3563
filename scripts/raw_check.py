"""N2 check: motion_correct_raw (fused conditioning) against condition_movie -> estimate_global_motion ->
motion_correct_sum on the same raw movie; timings of both routes."""
import sys, time, torch
sys.path.insert(0, ".")
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")


def raw_movie(t, h, w, dtype, seed, amp, pad=64):
    g = torch.Generator(device=dev).manual_seed(seed)
    base = torch.rand(h + 2 * pad, w + 2 * pad, generator=g, device=dev) * 40 + 10
    dy = torch.round(torch.linspace(-amp, amp + 2, t)).long().tolist()
    dx = torch.round(torch.linspace(amp - 1, -amp, t)).long().tolist()
    raw = torch.empty((t, h, w), dtype=dtype, device=dev)
    for f in range(t):
        v = base[pad - dy[f]: pad - dy[f] + h, pad - dx[f]: pad - dx[f] + w] + 6 * torch.randn(h, w, generator=g, device=dev)
        if dtype == torch.int16:
            v = v * 8 - 100  # wider range, negative values
            raw[f] = v.round().clamp(-32768, 32767).to(dtype)
        else:
            raw[f] = v.round().clamp(0, 255).to(dtype)
    gain = (1.0 + 0.1 * torch.randn(h, w, generator=g, device=dev)).clamp(0.5, 1.5)
    return raw, gain, dy, dx


def main():
  ok = True
  for (t, h, w, dtype, amp) in [(8, 4096, 4096, torch.uint8, 6), (8, 4096, 4096, torch.int16, 6), (10, 4096, 4096, torch.uint8, 24),
                                (6, 512, 512, torch.uint8, 4), (5, 300, 4096, torch.uint8, 3)]:
      raw, gain, dy, dx = raw_movie(t, h, w, dtype, 7, amp)
      img = mc.condition_movie(raw, gain)
      fa = mc.estimate_global_motion(img, 1.0)
      sa, fra = mc.motion_correct_sum(img, fa, 1.0, return_frames=True)
      fb, sb, frb = mc.motion_correct_raw(raw, gain, 1.0, return_frames=True)
      torch.cuda.synchronize()
      same = bool(torch.equal(fa, fb))
      es = float((sa - sb).abs().max() / sa.abs().max())
      ef = float((fra - frb).abs().max() / fra.abs().max())
      exp = torch.tensor([[dy[f] - dy[t // 2], dx[f] - dx[t // 2]] for f in range(t)], dtype=torch.float32)
      truth = bool(torch.equal(fa[:, :, 0, 0].T.cpu(), exp))
      print((t, h, w), dtype, f"amp {amp}: shifts equal {same} (== known drift {truth}), sum rel err {es:.2e}, frames rel err {ef:.2e}", flush=True)
      ok &= same and es < 1e-5 and ef < 1e-5
      del img, fra, frb

  t, h, w = 40, 4096, 4096
  raw, gain, _, _ = raw_movie(t, h, w, torch.uint8, 3, 7)
  def route_a():
      img = mc.condition_movie(raw, gain)
      f = mc.estimate_global_motion(img, 1.0)
      return mc.motion_correct_sum(img, f, 1.0, return_frames=True)
  def route_b():
      return mc.motion_correct_raw(raw, gain, 1.0, return_frames=True)
  for name, fn in (("condition_movie + fp32 path", route_a), ("fused raw path", route_b)):
      for _ in range(2): fn()
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(5): fn()
      torch.cuda.synchronize()
      print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per 40 x 4096^2 u8 movie (frames + sum out)", flush=True)
  torch.cuda.reset_peak_memory_stats(); base_mem = torch.cuda.memory_allocated()
  route_b(); torch.cuda.synchronize(); pb = torch.cuda.max_memory_allocated() - base_mem
  torch.cuda.reset_peak_memory_stats(); route_a(); torch.cuda.synchronize(); pa = torch.cuda.max_memory_allocated() - base_mem
  print(f"peak extra memory: fp32 route {pa / 1e9:.2f} GB, fused {pb / 1e9:.2f} GB")
  print("ALL OK" if ok else "MISMATCH")
  sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
