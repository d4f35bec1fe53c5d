"""Working counterpart of the reference's legacy ``model/pix2pix_model.py`` (class ``Pix2PixModel``).

The reference class is not runnable as shipped; this keeps its method
names and semantics (set_input, forward, backward_D with the 0.5 factor, backward_G,
optimize_parameters: reference model/pix2pix_model.py:100-154) on the HIP networks.
"""
import torch

from model import networks
from nirgan_hip.optim import HipAdam


class Pix2PixModel:
    def __init__(self, opt):
        cfg = opt
        opt = opt.base_configs
        self.opt, self.isTrain = opt, opt.isTrain
        self.loss_names = ['G_GAN', 'G_L1', 'D_real', 'D_fake']
        self.visual_names = ['real_A', 'fake_B', 'real_B']
        self.model_names = ['G', 'D'] if self.isTrain else ['G']
        if "satclip" in cfg and cfg.satclip.use_satclip:
            raise NotImplementedError("Pix2PixModel covers the plain generator; use Px2Px_PL for SatCLIP injection")
        print("Creating Standard Pix2Pix Generator.")
        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm,
                                      not opt.no_dropout, opt.init_type, opt.init_gain)
        self.optimizers = []
        if self.isTrain:
            self.netD = networks.define_D(opt.input_nc + opt.output_nc, opt.ndf, opt.netD,
                                          opt.n_layers_D, opt.norm, opt.init_type, opt.init_gain)
            self.criterionGAN = networks.GANLoss(opt.gan_mode)
            from model.pix2pix import HipL1Loss
            self.criterionL1 = HipL1Loss()
            self.optimizer_G = HipAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999), net=self.netG)
            self.optimizer_D = HipAdam(self.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999), net=self.netD)
            self.optimizers += [self.optimizer_G, self.optimizer_D]

    def to(self, device):
        self.netG.to(device)
        if self.isTrain:
            self.netD.to(device)
            self.criterionGAN.to(device)
        return self

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def set_input(self, input):
        AtoB = self.opt.direction == 'AtoB'
        self.real_A = input['A' if AtoB else 'B']
        self.real_B = input['B' if AtoB else 'A']

    def forward(self):
        self.fake_B = self.netG(self.real_A)

    def backward_D(self):
        fake_AB = torch.cat((self.real_A, self.fake_B), 1)
        pred_fake = self.netD(fake_AB.detach())
        self.loss_D_fake = self.criterionGAN(pred_fake, False)
        real_AB = torch.cat((self.real_A, self.real_B), 1)
        pred_real = self.netD(real_AB)
        self.loss_D_real = self.criterionGAN(pred_real, True)
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def backward_G(self):
        fake_AB = torch.cat((self.real_A, self.fake_B), 1)
        pred_fake = self.netD(fake_AB)
        self.loss_G_GAN = self.criterionGAN(pred_fake, True)
        self.loss_G_L1 = self.criterionL1(self.fake_B, self.real_B) * self.opt.lambda_L1
        self.loss_G = self.loss_G_GAN + self.loss_G_L1
        self.loss_G.backward()

    def optimize_parameters(self):
        self.forward()
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.step()
