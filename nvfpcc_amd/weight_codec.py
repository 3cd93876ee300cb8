"""Huffman coding of the 1/16-quantised decoder kernels -- the `net_weight_pack` entry of pack.pk
(/root/reference/util_code_quantized_weights.py:37-209), without the PyPI `bitstream` dependency.

Bit order: MSB first within a byte (numpy.packbits), stream padded with zero bits to a whole byte
(util_code_quantized_weights.py:119-126).  `bitstream` itself is not available offline, so this bit order is
the build's documented assumption; the stream LENGTH does not depend on it.
"""
import heapq

import numpy as np
import torch

qp = 16
keys_quantize = ['reconstructor.up0.kernel', 'reconstructor.conv0.kernel', 'reconstructor.up1.kernel',
                 'reconstructor.conv1.kernel', 'reconstructor.up2.kernel', 'reconstructor.conv2.kernel',
                 'reconstructor.conv2_cls.kernel']
keys_code_as_is = ['entropy_coder.sigma', 'entropy_coder.mu', 'reconstructor.activation.beta',
                   'reconstructor.activation.gamma', 'reconstructor.activation.pedestal', 'reconstructor.up0.b',
                   'reconstructor.conv0.b', 'reconstructor.up1.b', 'reconstructor.conv1.b', 'reconstructor.up2.b',
                   'reconstructor.conv2.b', 'reconstructor.conv2_cls.b', 'reconstructor.likelihood_model.sigma',
                   'reconstructor.likelihood_model.mu']


def read_elements_from_file(fn, qp_=qp):
    ws = torch.load(fn, map_location=torch.device('cpu'))
    pool = [ws[k].detach().numpy() * qp_ for k in keys_quantize]
    as_is_pool = [ws[k].detach().numpy() for k in keys_code_as_is]
    eles = np.concatenate([t.reshape(-1) for t in pool])
    if not np.abs(np.sum(np.round(eles) - eles)) < 1e-3:
        print("Warning: the loaded elements are not discrete!")
        raise ValueError('The loaded elements are not discrete.')
    return eles, pool, as_is_pool


def get_pdf(eles):
    """Relative frequency of every integer value that occurs, ascending."""
    vals = np.round(eles).astype(np.int64)
    lo, hi = int(np.min(eles)), int(np.max(eles))
    counts = np.bincount(vals - lo, minlength=hi - lo + 1).astype(np.float64)
    pdf = counts / counts.sum()
    nz = np.nonzero(pdf)[0]
    return pdf[nz], (nz + lo).astype(np.int64)


def get_huffman_codebook(pdf, bins):
    """Huffman tree; ties resolved like the reference's repeated stable sort (earlier-created node first,
    the first of the two merged nodes gets bit 0) so the codewords coincide, not only their lengths."""
    heap = [(float(p), i, (k, None)) for i, (p, k) in enumerate(zip(pdf, bins))]
    heapq.heapify(heap)
    counter = len(heap)
    while len(heap) > 1:
        pa, ia, a = heapq.heappop(heap)
        pb, ib, b = heapq.heappop(heap)
        heapq.heappush(heap, (pa + pb, counter, (None, (a, b))))
        counter += 1
    codebook, inv_codebook = {}, {}

    def walk(node, word):
        sym, kids = node
        if kids is None:
            codebook[sym] = np.array(word).astype(bool)
            inv_codebook[''.join(str(c) for c in word)] = sym
            return
        walk(kids[0], word + [0])
        walk(kids[1], word + [1])
    walk(heap[0][2], [])
    return codebook, inv_codebook


def est_rate(pdf, bins, codebook):
    return float(sum(len(codebook[s]) * p for p, s in zip(pdf, bins)))


def entropy_encode(tensor_list, codebook):
    shape_list = [t.shape for t in tensor_list]
    words = []
    for t in tensor_list:
        for v in t.reshape(-1):
            assert abs(int(v) - v) < 1e-3
            words.append(codebook[int(v)])
    bits = np.concatenate(words, 0) if words else np.zeros(0, bool)
    print("Length of the bit string: ", bits.shape)
    return np.packbits(bits.astype(np.uint8)).tobytes(), shape_list    # MSB first, zero-padded to a byte


def entropy_decode(byte_str, inv_codebook, nsymbol, shape_list):
    bits = np.unpackbits(np.frombuffer(byte_str, np.uint8))
    symbols, length, value = [], 0, 0
    if '' in inv_codebook:      # one distinct value: the empty codeword, zero bits per symbol
        symbols = [inv_codebook['']] * nsymbol
        bits = bits[:0]
    table = {(len(k), int(k, 2)): v for k, v in inv_codebook.items() if k}
    for b in bits:
        length += 1
        value = (value << 1) | int(b)
        hit = table.get((length, value))
        if hit is not None:
            symbols.append(hit)
            length, value = 0, 0
            if len(symbols) == nsymbol:
                break
    tensors, pos = [], 0
    for s in shape_list:
        n = int(np.prod(s))
        tensors.append(np.array(symbols[pos:pos + n]).astype(np.float32).reshape(s))
        pos += n
    return tensors


def est_fp_bit_consumption(tensor_list):
    return sum(int(np.prod(t.shape)) for t in tensor_list) * 32


def enc_dec_from_file(filename, qp=qp):
    eles, pool, as_is_pool = read_elements_from_file(filename, qp)
    pdf, bins = get_pdf(eles)
    codebook, inv_codebook = get_huffman_codebook(pdf, bins)
    print('Estimated E(l): ', est_rate(pdf, bins, codebook))
    bit_stream, shape_list = entropy_encode(pool, codebook)
    print('Bit-stream length in bytes: ', len(bit_stream))
    n_bits_as_is = est_fp_bit_consumption(as_is_pool)
    print('Extra bits: ', n_bits_as_is)
    print('Total bits: ', n_bits_as_is + len(bit_stream) * 8)
    dec_pool = entropy_decode(bit_stream, inv_codebook, len(eles), shape_list)
    for a, b in zip(pool, dec_pool):
        assert np.sum(np.abs(a - b)) < 1e-6
    return {'bit_stream': bit_stream, 'inv_codebook': inv_codebook, 'element_length': len(eles),
            'shape_list': shape_list, 'as_is_pool': as_is_pool, 'keys_quantize': keys_quantize,
            'keys_code_as_is': keys_code_as_is}
