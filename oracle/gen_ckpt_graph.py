"""Pins the transform GRAPH to what the reference itself holds: TensorFlow is absent and no
trained tensors are in the mount, but every `kodak_tensorflow/eae/results/*/model_*.ckpt.meta` is a MetaGraphDef, i.e. the
complete op graph `EntropyAutoencoder` built -- which ops, in which order, wired to which variables, with which strides /
paddings / data formats / filter shapes / output shapes. This script walks that graph from the input placeholder to the
reconstruction (no TensorFlow needed: kodak/eae/graph/tf_checkpoint.py reads the protocol buffers) and writes the forward
path as data:

    tests/golden/ckpt_graph.json   {model file: {"placeholder": ..., "layers": [ {kind, node, attrs, variables, ...} ]}}

`tests/test_oracle_graph.py` then asserts that oracle/transforms.py (`ENCODER_LAYERS`, `DECODER_LAYERS`, the tables its
`encoder` / `decoder` interpret) is exactly that path: same op kinds in the same order, same variables, strides, paddings,
filter layouts, and the GDN op chain Reshape -> Square -> MatMul(x^2, gamma) -> Add(beta) -> Sqrt -> Div (Mul for the
inverse) of tfutils.py:393-397, 505-509. Run here (needs /root/reference): `python oracle/gen_ckpt_graph.py`.
"""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoencoder_based_image_compression_amd.kodak.eae.graph import tf_checkpoint as ckpt  # noqa: E402

RESULTS = '/root/reference/kodak_tensorflow/eae/results'
DT_FLOAT, DT_INT32 = 1, 3


def packed_varints(buffer):
    (out, position) = ([], 0)
    buffer = memoryview(buffer)
    while position < len(buffer):
        (value, position) = ckpt.read_varint(buffer, position)
        out.append(value)
    return out


def parse_attr(raw):
    """AttrValue -> python: s (2) / i (3) / f (4) / b (5) / type (6) / shape (7) / tensor (8) / list (1)."""
    out = None
    for (n, w, v) in ckpt.iterate_fields(raw):
        if n == 2:
            out = bytes(v).decode('latin-1')
        elif n == 3:
            out = v if v < (1 << 63) else v - (1 << 64)
        elif n == 4:
            out = struct.unpack('<f', struct.pack('<I', v))[0] if isinstance(v, int) else struct.unpack('<f', bytes(v))[0]
        elif n == 5:
            out = bool(v)
        elif n == 6:
            out = {'type': v}
        elif n == 7:
            out = {'shape': list(ckpt.parse_tensor_shape(v))}
        elif n == 8:
            out = parse_tensor(v)
        elif n == 1:
            ints = []
            for (n2, w2, v2) in ckpt.iterate_fields(v):
                if n2 == 3:                                   # ListValue.i: packed or repeated varints
                    if w2 == 2:
                        ints.extend(packed_varints(v2))
                    else:
                        ints.append(v2)
            out = ints
    return out


def parse_tensor(raw):
    """TensorProto -> {'dtype', 'shape', 'values'} for small int32 / float tensors (shapes, scalars)."""
    (dtype, shape, content, ints, floats) = (None, [], None, [], [])
    for (n, w, v) in ckpt.iterate_fields(raw):
        if n == 1:
            dtype = v
        elif n == 2:
            shape = list(ckpt.parse_tensor_shape(v))
        elif n == 4:
            content = bytes(v)
        elif n == 7:                                          # int_val
            if w == 2:
                ints.extend(packed_varints(v))
            else:
                ints.append(v)
        elif n == 5:                                          # float_val
            if w == 2:
                floats.extend(struct.unpack('<{}f'.format(len(v)//4), bytes(v)))
            else:
                floats.append(struct.unpack('<f', struct.pack('<I', v))[0])
    values = None
    count = 1
    for d in shape:
        count *= d
    if dtype == DT_INT32:
        if content is not None:
            values = list(struct.unpack('<{}i'.format(len(content)//4), content))
        else:
            values = [x if x < (1 << 31) else x - (1 << 32) for x in (x & 0xFFFFFFFF for x in ints)]
            if len(values) == 1 and count > 1:
                values = values*count
    elif dtype == DT_FLOAT and count <= 16:
        if content is not None:
            values = list(struct.unpack('<{}f'.format(len(content)//4), content))
        else:
            values = list(floats)
    return {'dtype': dtype, 'shape': shape, 'values': values}


def read_graph(path):
    with open(path, 'rb') as file:
        meta = file.read()
    nodes = {}
    for (number, wire, value) in ckpt.iterate_fields(meta):
        if number != 2 or wire != 2:                                    # MetaGraphDef.graph_def
            continue
        for (n1, w1, node) in ckpt.iterate_fields(value):
            if n1 != 1 or w1 != 2:                                      # GraphDef.node
                continue
            (name, op, inputs, attrs) = (None, None, [], {})
            for (n2, w2, v2) in ckpt.iterate_fields(node):
                if n2 == 1:
                    name = bytes(v2).decode()
                elif n2 == 2:
                    op = bytes(v2).decode()
                elif n2 == 3:
                    inputs.append(bytes(v2).decode())
                elif n2 == 5:
                    (key, attr) = (None, None)
                    for (n3, w3, v3) in ckpt.iterate_fields(v2):
                        if n3 == 1:
                            key = bytes(v3).decode()
                        elif n3 == 2:
                            attr = bytes(v3)
                    attrs[key] = attr
            nodes[name] = {'op': op, 'inputs': [i for i in inputs if not i.startswith('^')], 'attrs': attrs}
    return nodes


class Walker(object):
    def __init__(self, nodes):
        self.nodes = nodes
        self.consumers = {}
        for (name, node) in nodes.items():
            if name.startswith(('gradients', 'save', 'Adam')):
                continue
            for i in node['inputs']:
                self.consumers.setdefault(i.split(':')[0], []).append(name)

    def op(self, name):
        return self.nodes[name.split(':')[0]]['op']

    def inputs(self, name):
        return [i.split(':')[0] for i in self.nodes[name]['inputs']]

    def variable(self, name):
        """Follows `<var>/read` Identity nodes to the Variable they read."""
        name = name.split(':')[0]
        while self.op(name) == 'Identity':
            name = self.inputs(name)[0]
        assert self.op(name) in ('Variable', 'VariableV2'), (name, self.op(name))
        return name

    def variable_shape(self, name):
        return parse_attr(self.nodes[name]['attrs']['shape'])['shape']

    def attr(self, name, key):
        return parse_attr(self.nodes[name]['attrs'][key])

    def const(self, name):
        node = self.nodes[name.split(':')[0]]
        if node['op'] == 'Const':
            return parse_attr(node['attrs']['value'])['values']
        if node['op'] == 'Pack':                                        # tf.stack of scalars
            return [self.const(i)[0] if self.const(i) is not None else None for i in self.inputs(name.split(':')[0])]
        return None

    def only_consumer(self, name, ops):
        found = [c for c in self.consumers.get(name, []) if self.op(c) in ops]
        assert len(found) == 1, (name, ops, self.consumers.get(name))
        return found[0]

    def gdn_after(self, x):
        """The op chain tfuls.gdn / inverse_gdn builds on tensor `x` (tfutils.py:393-397, 505-509), or None."""
        reshapes = [c for c in self.consumers.get(x, []) if self.op(c) == 'Reshape'
                    and any(self.op(cc) == 'Square' for cc in self.consumers.get(c, []))]
        if len(reshapes) != 1:                                          # other Reshape consumers (loss terms) do not square
            return None
        reshape = reshapes[0]
        squares = [c for c in self.consumers.get(reshape, []) if self.op(c) == 'Square']
        if len(squares) != 1:
            return None
        square = squares[0]
        matmul = self.only_consumer(square, ('MatMul',))
        add = self.only_consumer(matmul, ('Add',))
        sqrt = self.only_consumer(add, ('Sqrt',))
        last = self.only_consumer(sqrt, ('Div', 'RealDiv', 'Mul'))
        back = self.only_consumer(last, ('Reshape',))
        mm_in = self.inputs(matmul)
        add_in = self.inputs(add)
        last_in = self.inputs(last)
        # beta enters as tf.tile(tf.reshape(beta, [1, C]), [rows, 1]) (tfutils.py:396-397)
        beta_ops = []
        beta_node = add_in[1]
        while self.op(beta_node) in ('Tile', 'Reshape'):
            beta_ops.append(self.op(beta_node))
            beta_node = self.inputs(beta_node)[0]
        return {
            'kind': 'inverse_gdn' if self.op(last) == 'Mul' else 'gdn',
            'ops': [self.op(n) for n in (reshape, square, matmul, add, sqrt, last, back)],
            'nodes': [reshape, square, matmul, add, sqrt, last, back],
            'reshape_to': self.const(self.inputs(reshape)[1]),
            'matmul': {'a_is_the_square': mm_in[0] == square, 'b': self.variable(mm_in[1]),
                       'transpose_a': self.attr(matmul, 'transpose_a'), 'transpose_b': self.attr(matmul, 'transpose_b')},
            'add': {'first_is_the_matmul': add_in[0] == matmul, 'second_via': beta_ops, 'second': self.variable(beta_node)},
            'last': {'first_is_the_reshaped_input': last_in[0] == reshape, 'second_is_the_sqrt': last_in[1] == sqrt},
            'gamma': self.variable(mm_in[1]), 'gamma_shape': self.variable_shape(self.variable(mm_in[1])),
            'beta': self.variable(beta_node), 'beta_shape': self.variable_shape(self.variable(beta_node)),
            'output': back,
        }

    def conv_layer(self, name):
        node = self.nodes[name]
        transposed = node['op'] == 'Conv2DBackpropInput'
        ins = self.inputs(name)
        (data, filt) = (ins[2], ins[1]) if transposed else (ins[0], ins[1])
        layer = {
            'kind': 'conv2d_transpose' if transposed else 'conv2d', 'node': name, 'input': data,
            'filter': self.variable(filt), 'filter_shape': self.variable_shape(self.variable(filt)),
            'strides': self.attr(name, 'strides'), 'padding': self.attr(name, 'padding'),
            'data_format': self.attr(name, 'data_format'), 'bias': None, 'output': name,
        }
        if transposed:
            layer['output_shape'] = self.const(ins[0])
        biases = [c for c in self.consumers.get(name, []) if self.op(c) == 'BiasAdd']
        if biases:
            assert len(biases) == 1
            layer['bias'] = self.variable(self.inputs(biases[0])[1])
            layer['bias_data_format'] = self.attr(biases[0], 'data_format')
            layer['bias_node'] = biases[0]
            layer['output'] = biases[0]
        return layer


def forward_path(path):
    nodes = read_graph(path)
    walker = Walker(nodes)
    forward = [n for n in nodes if not n.startswith(('gradients', 'save', 'Adam')) and nodes[n]['op'] in ('Conv2D', 'Conv2DBackpropInput')]
    # GraphDef order is construction order: encoder convolutions first, decoder ones after
    layers = []
    placeholder = None
    for name in forward:
        layer = walker.conv_layer(name)
        producer = layer['input']
        if walker.op(producer) == 'Placeholder':
            placeholder = {'node': producer, 'dtype_enum': walker.attr(producer, 'dtype')['type'], 'shape': walker.attr(producer, 'shape')['shape']}
        # a normalisation in FRONT of this convolution that no earlier layer accounts for (inverse_gdn #4 on the latents)
        if layers and producer != layers[-1]['output'] and walker.op(producer) == 'Reshape':
            chain_last = walker.inputs(producer)[0]
            if walker.op(chain_last) in ('Mul', 'Div', 'RealDiv'):
                reshaped = walker.inputs(chain_last)[0]
                source = walker.inputs(reshaped)[0]
                norm = walker.gdn_after(source)
                if norm is not None and norm['output'] == producer and (not layers or layers[-1].get('output') != producer):
                    norm['input'] = source
                    layers.append(norm)
        layers.append(layer)
        norm = walker.gdn_after(layer['output'])
        if norm is not None:
            norm['input'] = layer['output']
            layers.append(norm)
    return {'placeholder': placeholder, 'layers': layers}


def main():
    out = {}
    for suffix in sorted(os.listdir(RESULTS)):
        for name in sorted(os.listdir(os.path.join(RESULTS, suffix))):
            if name.endswith('.ckpt.meta'):
                out['{0}/{1}'.format(suffix, name)] = forward_path(os.path.join(RESULTS, suffix, name))
    path = os.path.join(ROOT, 'tests', 'golden', 'ckpt_graph.json')
    with open(path, 'w') as file:
        json.dump(out, file, indent=1, sort_keys=True)
    for (key, value) in out.items():
        print(key, [layer['kind'] for layer in value['layers']])


if __name__ == '__main__':
    main()
