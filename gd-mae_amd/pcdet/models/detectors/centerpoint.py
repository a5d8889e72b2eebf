"""``CenterPoint`` detector wrapper of the GD-MAE fine-tune configs: module loop + ``dense_head.get_loss()``
(reference pcdet/models/detectors/centerpoint.py:4-35).  ``sync_loss_scalar`` as in ``GDMAE``: the reference reads
``loss_rpn.item()`` every step; set it to False to keep the step free of host syncs."""
from .detector3d_template import Detector3DTemplate


class CenterPoint(Detector3DTemplate):
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
        """(pred_dicts, recall_dict) as the reference's CenterPoint.post_processing (centerpoint.py:37-52)."""
        cfg = self.model_cfg.POST_PROCESSING
        final = batch_dict['final_box_dicts']
        recall = {}
        for index in range(batch_dict['batch_size']):
            recall = self.generate_recall_record(box_preds=final[index]['pred_boxes'], recall_dict=recall, batch_index=index,
                                                 data_dict=batch_dict, thresh_list=cfg.RECALL_THRESH_LIST)
        return final, recall

    def get_training_loss(self):
        loss_rpn, tb_dict = self.dense_head.get_loss()
        tb_dict = {'loss_rpn': loss_rpn.item() if self.sync_loss_scalar else loss_rpn.detach(), **tb_dict}
        return loss_rpn, tb_dict, {}
