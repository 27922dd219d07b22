"""Plot helper of the reference package (loop_closure_detection/utils.py:6-33); visualisation only, off the hot
path.  matplotlib is imported on use."""
from typing import Optional


def plot_image_matches(image_0, image_1, image_id_0: Optional[int] = None, image_id_1: Optional[int] = None,
                       cosine_similarity: Optional[float] = None, save_figure: bool = True) -> None:
    import matplotlib.pyplot as plt
    fig = plt.figure()
    for pos, img, title in ((211, image_0, image_id_0), (212, image_1, image_id_1)):
        plt.subplot(pos)
        plt.imshow(img)
        plt.axis('off')
        if title is not None:
            plt.title(title)
    if cosine_similarity is not None:
        plt.suptitle(f'cos_sim = {cosine_similarity}')
    if save_figure:
        assert image_id_0 is not None and image_id_1 is not None
        plt.savefig(f'./figures/sequence_08/matches/{image_id_0:04}_{image_id_1:04}.png')
    else:
        plt.show()
    plt.close(fig)
