// pcl/registration/correspondence_rejection_trimmed.h - see correspondence_rejection.h
#include <pcl/registration/correspondence_rejection.h>
