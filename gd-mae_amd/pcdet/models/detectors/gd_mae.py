"""``GDMAE`` detector wrapper (reference pcdet/models/detectors/gd_mae.py:4-37): run the module list,
return ``({'loss'}, tb_dict, disp_dict)`` in training.  ``tb_dict['loss_rpn']`` is kept as a 0-dim tensor
unless ``sync_loss_scalar`` is set - the reference's ``.item()`` forces a host sync every step."""
from .detector3d_template import Detector3DTemplate


class GDMAE(Detector3DTemplate):
    sync_loss_scalar = True

    def __init__(self, model_cfg, num_class, dataset, logger):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        for m in self.module_list:
            batch_dict = m(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def post_processing(self, batch_dict):
        return {}, {}

    def get_training_loss(self):
        loss_rpn, tb_dict = self.backbone_3d.get_loss()
        tb_dict = {'loss_rpn': loss_rpn.item() if self.sync_loss_scalar else loss_rpn.detach(), **tb_dict}
        return loss_rpn, tb_dict, {}
