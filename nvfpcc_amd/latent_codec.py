"""Arithmetic coding of the integer latents (encode(): NVFPCC.py:446-482, decode(): :586-607) through the
in-process range coder libnvf_codec.so (include/nvf_codec.h) instead of `./module_arithmeticcoding` over pipes.
The `latent_pack` dict has the reference's keys and value types."""
import ctypes as C
import os

import numpy as np
import torch

OFFSET = 512          # symbols = rounded latent + 512, mu + 512 (NVFPCC.py:447-458)
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnvf_codec.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: run python -m nvfpcc_amd.build")
        h = C.CDLL(path)
        h.nvf_ac_encode.restype = C.c_int64
        h.nvf_ac_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int64]
        h.nvf_ac_decode.restype = C.c_int
        h.nvf_ac_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                    C.c_void_p]
        h.nvf_codec_version.restype = C.c_int
        _LIB = h
    return _LIB


def encode_symbols(symbols, mu, sigma, level_mu=1, level_sigma=1):
    """int16 symbols in [0,1023], float32 mu/sigma per symbol -> bytes."""
    symbols = np.ascontiguousarray(symbols, np.int16)
    mu = np.ascontiguousarray(mu, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    n = symbols.shape[0]
    assert mu.shape[0] == n and sigma.shape[0] == n
    out = np.empty(8 * n + 64, np.uint8)
    r = lib().nvf_ac_encode(symbols.ctypes.data, mu.ctypes.data, sigma.ctypes.data, n, level_mu, level_sigma,
                            out.ctypes.data, out.shape[0])
    if r < 0:
        raise ValueError({-1: "symbol out of range or with zero frequency", -2: "output buffer too small"}[int(r)])
    return out[:r].tobytes()


def decode_symbols(stream, mu, sigma, level_mu=1, level_sigma=1):
    mu = np.ascontiguousarray(mu, np.float32)
    sigma = np.ascontiguousarray(sigma, np.float32)
    n = mu.shape[0]
    buf = np.frombuffer(stream, np.uint8)
    out = np.empty(n, np.int16)
    r = lib().nvf_ac_decode(buf.ctypes.data if buf.size else None, buf.size, mu.ctypes.data, sigma.ctypes.data, n,
                            level_mu, level_sigma, out.ctypes.data)
    if r != 0:
        raise ValueError("latent stream is inconsistent with the (mu, sigma) model")
    return out


def _tiled(param, shape):
    return torch.tile(param, (shape[0], 1, shape[2], shape[3], shape[4])).detach().cpu().numpy().astype(
        np.float32).reshape(-1)


def arithmetic_enc(tensor, sigma, mu):
    """quantised latents [N,ch,2,2,2] + abs(sigma), mu [1,ch,1,1,1] -> latent_pack (with the reference's
    encode-then-decode self-check, NVFPCC.py:466-470)."""
    s = tensor.shape
    np_tensor = tensor.detach().cpu().numpy()
    d = np_tensor.astype(np.int16)
    assert np.sum(np.abs(d - np_tensor)) < 1e-6, "latents must be integers"
    flat_coeff = d.reshape(-1) + OFFSET
    flat_sigma = _tiled(sigma, s)
    flat_mu = _tiled(mu, s) + OFFSET
    stream = encode_symbols(flat_coeff, flat_mu, flat_sigma)
    print('Latent code byte-stream length: ', len(stream))
    back = decode_symbols(stream, flat_mu, flat_sigma).astype(np.float32) - OFFSET
    assert np.sum(np.abs(back.reshape(s) - np_tensor)) < 1e-6
    return {'shape': s, 'latent_byte_stream': stream, 'sigma': sigma, 'mu': mu,
            'length': np.array([flat_coeff.shape[0]], dtype=np.int64)}


def arithmetic_dec(latent_pack):
    s = latent_pack['shape']
    flat_sigma = _tiled(latent_pack['sigma'], s)
    flat_mu = _tiled(latent_pack['mu'], s) + OFFSET
    print('Latent code byte-stream length: ', len(latent_pack['latent_byte_stream']))
    sym = decode_symbols(latent_pack['latent_byte_stream'], flat_mu, flat_sigma)
    return torch.from_numpy((sym.astype(np.float32) - OFFSET).reshape(tuple(s)))
