"""ORACLE — test infrastructure only.  CPU restatement of the reference's conditional WGAN-GP,
`conditional_gan/mnist/mnist_wgan_conditional.py`:
    Hyperparameter :21-31, Generator :51-78, Critic :80-108, optimizers :118-119,
    critic update :133-155 (gradient penalty :146-150), generator update :157-168
on the same PyTorch operators (the gradient penalty's second-order terms come from torch autograd with create_graph=True,
exactly as the reference obtains them).  The reference file cannot be imported (torchvision, a CUDA device query and the
dataset at import, training at import): tests/golden/make_golden.py lifts its class definitions and loop body out of the
syntax tree, runs them on the CPU at reduced width and records the draws they made, so this restatement is pinned on them.
"""
from dataclasses import dataclass

import torch
from torch import autograd, nn, optim


@dataclass
class Hyperparameter:                      # :21-31
    num_classes: int = 10
    batchsize: int = 128
    latent_size: int = 32
    n_critic: int = 5
    critic_size: int = 1024
    generator_size: int = 1024
    critic_hidden_size: int = 1024
    gp_lambda: float = 10.0


class Generator(nn.Module):
    def __init__(self, hp):
        super().__init__()
        gs = hp.generator_size
        self.hp = hp
        self.latent_embedding = nn.Sequential(nn.Linear(hp.latent_size, gs // 2))                 # :54-56
        self.condition_embedding = nn.Sequential(nn.Linear(hp.num_classes, gs // 2))             # :57-59
        self.tcnn = nn.Sequential(                                                               # :60-71
            nn.ConvTranspose2d(gs, gs, 4, 1, 0), nn.BatchNorm2d(gs), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(gs, gs // 2, 3, 2, 1), nn.BatchNorm2d(gs // 2), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(gs // 2, gs // 4, 4, 2, 1), nn.BatchNorm2d(gs // 4), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(gs // 4, 1, 4, 2, 1), nn.Tanh())

    def forward(self, latent, condition):
        vec_latent = self.latent_embedding(latent)
        vec_class = self.condition_embedding(condition)
        combined = torch.cat([vec_latent, vec_class], dim=1).reshape(-1, self.hp.generator_size, 1, 1)   # :76
        return self.tcnn(combined)


class Critic(nn.Module):
    def __init__(self, hp):
        super().__init__()
        cs = hp.critic_size
        self.condition_embedding = nn.Sequential(nn.Linear(hp.num_classes, cs * 4))              # :83-85
        self.cnn_net = nn.Sequential(                                                            # :86-97
            nn.Conv2d(1, cs // 4, 3, 2), nn.InstanceNorm2d(cs // 4, affine=True), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(cs // 4, cs // 2, 3, 2), nn.InstanceNorm2d(cs // 2, affine=True), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(cs // 2, cs, 3, 2), nn.InstanceNorm2d(cs, affine=True), nn.LeakyReLU(0.2, inplace=True),
            nn.Flatten())
        self.Critic_net = nn.Sequential(                                                         # :98-102
            nn.Linear(cs * 8, hp.critic_hidden_size), nn.LeakyReLU(0.2, inplace=True), nn.Linear(hp.critic_hidden_size, 1))

    def forward(self, image, condition):
        vec_condition = self.condition_embedding(condition)
        cnn_features = self.cnn_net(image)
        combined = torch.cat([cnn_features, vec_condition], dim=1)                                # :107
        return self.Critic_net(combined)


def make_optimizers(critic, generator):
    return (optim.AdamW(critic.parameters(), lr=1e-4, betas=(0.0, 0.9)),                         # :118
            optim.AdamW(generator.parameters(), lr=1e-4, betas=(0.0, 0.9)))                      # :119


def critic_step(critic, generator, critic_optimizer, hp, real_images, real_class_labels, noise, alpha):
    """:133-155 with the draws (noise :141, alpha :146) supplied.  real_class_labels: one-hot rows (all_labels[idx], :133)."""
    bs = real_images.shape[0]
    grad_tensor = torch.ones((bs, 1), dtype=real_images.dtype)                                   # :126
    critic_optimizer.zero_grad()
    critic_output_real = critic(real_images, real_class_labels)
    critic_loss_real = critic_output_real.mean()
    with torch.no_grad():
        fake_image = generator(noise, real_class_labels)                                         # :142 (train-mode BatchNorm)
    critic_output_fake = critic(fake_image, real_class_labels)
    critic_loss_fake = critic_output_fake.mean()
    interpolates = (alpha.view(-1, 1, 1, 1) * real_images + ((1. - alpha.view(-1, 1, 1, 1)) * fake_image)).requires_grad_(True)
    d_interpolates = critic(interpolates, real_class_labels)
    gradients = autograd.grad(d_interpolates, interpolates, grad_tensor, create_graph=True, only_inputs=True)[0]   # :149
    gradient_penalty = hp.gp_lambda * ((gradients.view(bs, -1).norm(dim=1) - 1.) ** 2).mean()   # :150
    critic_loss = -critic_loss_real + critic_loss_fake + gradient_penalty                        # :152
    critic_loss.backward()
    critic_optimizer.step()
    return {"critic_loss": critic_loss.item(), "loss_real": critic_loss_real.item(), "loss_fake": critic_loss_fake.item(),
            "gradient_penalty": gradient_penalty.item(), "gradients": gradients.detach(), "fake_image": fake_image}


def generator_step(critic, generator, generator_optimizer, fake_class_labels, noise):
    """:157-168 with the draws (labels :161, noise :162) supplied."""
    generator_optimizer.zero_grad()
    fake_image = generator(noise, fake_class_labels)
    critic_output_fake = critic(fake_image, fake_class_labels)
    generator_loss = -critic_output_fake.mean()
    generator_loss.backward()
    generator_optimizer.step()
    return {"generator_loss": generator_loss.item()}


def build(hp, seed=1):
    torch.manual_seed(seed)                                                                      # :13
    critic, generator = Critic(hp), Generator(hp)                                                # :116 (critic first)
    return critic, generator


def synthetic_batch(hp, batch, seed, dtype=torch.float32):
    """x ~ U[-1,1) [B,1,28,28]; one-hot labels; z ~ N(0,1) [B,latent]; alpha ~ U[0,1) [B,1]; and the
    generator step's own labels and noise."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 1, 28, 28, generator=g, dtype=dtype) * 2 - 1
    eye = torch.eye(hp.num_classes, dtype=dtype)
    y = eye[torch.randint(0, hp.num_classes, (batch,), generator=g)]
    z = torch.randn(batch, hp.latent_size, generator=g, dtype=dtype)
    alpha = torch.rand(batch, 1, generator=g, dtype=dtype)
    y2 = eye[torch.randint(0, hp.num_classes, (batch,), generator=g)]
    z2 = torch.randn(batch, hp.latent_size, generator=g, dtype=dtype)
    return x, y, z, alpha, y2, z2
