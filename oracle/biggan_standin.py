"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BigGAN-deep generator (BASELINE.json config 5).

**Parity unpinned**: the mounted reference snapshot contains no BigGAN source.
This follows the published architecture (Brock et al., ICLR 2019, appendix B) with the semantics of the
authors' PyTorch release (`BigGANdeep.Generator.forward`, `layers.GBlock`, `layers.ccbn`,
`layers.Attention`), spectral norm assumed folded into the weights:
  * y = cat([shared(labels), z], 1) conditions every cBN; h = linear(y).view(B, C, 4, 4)
  * GBlock: h = conv1(relu(bn1(x, y))); h = relu(bn2(h, y)); x = x[:, :Cout] if Cin != Cout;
            if upsample: h, x = up2(h), up2(x); h = conv2(h); h = conv3(relu(bn3(h, y)));
            h = conv4(relu(bn4(h, y))); return h + x
  * ccbn(x, y) = F.batch_norm(x, stored_mean, stored_var, eps) * (1 + gain(y)) + bias(y)
  * Attention: theta = theta(x); phi = maxpool2(phi(x)); g = maxpool2(g(x));
               beta = softmax(theta^T phi, -1); o = o(g beta^T); return gamma * o + x
  * output: tanh(conv3x3(relu(bn(h)))) with a plain BN (gain/bias parameters, stored statistics)
Functional, driven by a state_dict with that release's key names (the ones pretorched_x_amd.biggan uses).
"""
import torch
import torch.nn.functional as F


def ccbn(sd, x, y, p, eps):
    gain = 1 + F.linear(y, sd[p + ".gain.weight"])
    bias = F.linear(y, sd[p + ".bias.weight"])
    out = F.batch_norm(x, sd[p + ".stored_mean"], sd[p + ".stored_var"], None, None, False, 0.1, eps)
    return out * gain[:, :, None, None] + bias[:, :, None, None]


def conv(sd, x, p, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), 1, padding)


def gblock(sd, x, y, p, eps, upsample):
    cout = sd[p + ".conv4.weight"].shape[0]
    h = conv(sd, F.relu(ccbn(sd, x, y, p + ".bn1", eps)), p + ".conv1")
    h = F.relu(ccbn(sd, h, y, p + ".bn2", eps))
    if x.shape[1] != cout:
        x = x[:, :cout]
    if upsample:
        h = F.interpolate(h, scale_factor=2)
        x = F.interpolate(x, scale_factor=2)
    h = conv(sd, h, p + ".conv2", 1)
    h = conv(sd, F.relu(ccbn(sd, h, y, p + ".bn3", eps)), p + ".conv3", 1)
    h = conv(sd, F.relu(ccbn(sd, h, y, p + ".bn4", eps)), p + ".conv4")
    return h + x


def attention(sd, x, p):
    b, ch, hh, ww = x.shape
    theta = conv(sd, x, p + ".theta").view(b, ch // 8, hh * ww)
    phi = F.max_pool2d(conv(sd, x, p + ".phi"), 2).view(b, ch // 8, hh * ww // 4)
    g = F.max_pool2d(conv(sd, x, p + ".g"), 2).view(b, ch // 2, hh * ww // 4)
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
    o = conv(sd, torch.bmm(g, beta.transpose(1, 2)).view(b, ch // 2, hh, ww), p + ".o")
    return sd[p + ".gamma"] * o + x


def pre_tanh(sd, z, yemb, depth=2, bottom_width=4, eps=1e-5):
    with torch.no_grad():
        y = torch.cat([yemb, z], 1)
        h = F.linear(y, sd["linear.weight"], sd["linear.bias"]).view(z.size(0), -1, bottom_width, bottom_width)
        i = 0
        while ("blocks.%d.0.conv1.weight" % i) in sd:
            for j in range(depth):
                h = gblock(sd, h, y, "blocks.%d.%d" % (i, j), eps, upsample=(j == depth - 1))
            if ("blocks.%d.%d.theta.weight" % (i, depth)) in sd:
                h = attention(sd, h, "blocks.%d.%d" % (i, depth))
            i += 1
        p = "output_layer.0"
        h = F.batch_norm(h, sd[p + ".stored_mean"], sd[p + ".stored_var"], sd[p + ".gain"], sd[p + ".bias"], False, 0.1, eps)
        return conv(sd, F.relu(h), "output_layer.2", 1)


def forward(sd, z, yemb, **kw):
    return torch.tanh(pre_tanh(sd, z, yemb, **kw))
